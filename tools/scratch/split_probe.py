#!/usr/bin/env python3
"""Split-operand (3 x bf16 planes) policy layers against the exact-fp32 MFMA layer kernel: error against the float64 product and
time per launch, both networks per launch, on the PPO policy's three hidden-layer shapes at 4096 rows.

    python tools/scratch/split_probe.py [--device cpu] [--rows 4096] [--iters 50]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from massive_marl_benchmark_amd import _lib  # noqa: E402


def ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def p32_bytes(rows, K):
    return rows * ((K + 31) // 32) * 192


def planes_to_f32(planes, rows, K):
    """P32 bytes -> [rows, K] float32 (exact: the three planes sum to the fp32 number)"""
    KC = (K + 31) // 32
    v = planes.view(torch.bfloat16).view(rows, KC, 3, 32).float()
    return ((v[:, :, 0] + v[:, :, 1]) + v[:, :, 2]).reshape(rows, KC * 32)[:, :K]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device(args.device)
    L, d, stream = _lib.for_device(dev)
    M = args.rows
    torch.manual_seed(0)
    for (K, N, out_planes) in ((388, 1024, 1), (1024, 1024, 1), (1024, 512, 0)):
        xs, ws, bs = [], [], []
        for g in range(2):
            x = torch.randn(M, K, device=dev)
            xs.append(torch.where(x > 0, x, torch.expm1(x)).contiguous())
            ws.append((torch.randn(N, K, device=dev) / K ** 0.5).contiguous())
            bs.append((torch.randn(N, device=dev) * 0.1).contiguous())
        ref = [torch.nn.functional.elu(xs[g].double() @ ws[g].double().t() + bs[g].double()) for g in range(2)]
        rms = float(torch.cat(ref).pow(2).mean().sqrt())
        # exact-fp32 MFMA kernel
        y32 = [torch.empty(M, N, device=dev) for _ in range(2)]

        def run32():
            _lib.check(L.mms_linear2_act(d, M, N, K, xs[0].data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), y32[0].data_ptr(), xs[1].data_ptr(),
                                         ws[1].data_ptr(), bs[1].data_ptr(), y32[1].data_ptr(), 1, stream), what="linear2", L=L)
        # split kernel
        xp = [torch.empty(p32_bytes(M, K), dtype=torch.uint8, device=dev) for _ in range(2)]
        wp = [torch.empty(p32_bytes(N, K), dtype=torch.uint8, device=dev) for _ in range(2)]
        for g in range(2):
            _lib.check(L.mms_split_planes(d, M, K, 0, xs[g].data_ptr(), xp[g].data_ptr(), stream), what="split x", L=L)
            _lib.check(L.mms_split_planes(d, N, K, 0, ws[g].data_ptr(), wp[g].data_ptr(), stream), what="split w", L=L)
            back = planes_to_f32(xp[g], M, K)
            assert torch.equal(back, xs[g]), "planes do not sum back to the fp32 input"
        ysp = [torch.empty(p32_bytes(M, N) if out_planes else M * N * 4, dtype=torch.uint8, device=dev) for _ in range(2)]
        px, pw, pb, py = ptrs(xp), ptrs(wp), ptrs(bs), ptrs(ysp)

        def runsp():
            _lib.check(L.mms_linear_group_act_split(d, 2, M, N, K, px, pw, pb, py, 1, out_planes, None, None, None, None, None, 0, stream), what="split layer", L=L)
        run32()
        runsp()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        out = [planes_to_f32(ysp[g], M, N) if out_planes else ysp[g].view(torch.float32).view(M, N) for g in range(2)]
        for name, ys in (("fp32 mfma", y32), ("split bf16x3", out)):
            e = torch.cat([(ys[g].double() - ref[g]) for g in range(2)])
            print("K %4d N %4d  %-13s max err %.3e  rms err %.3e  mean err %+.2e  (units of rms(Y) = %.3f)" %
                  (K, N, name, float(e.abs().max()) / rms, float(e.pow(2).mean().sqrt()) / rms, float(e.mean()) / rms, rms), flush=True)
        if dev.type == "cuda":
            for name, fn in (("fp32 mfma", run32), ("split bf16x3", runsp)):
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                print("K %4d N %4d  %-13s %.1f us per launch (both networks) = %.1f TFLOP/s fp32-equivalent" %
                      (K, N, name, us, 2 * 2.0 * M * N * K / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
