#!/usr/bin/env python3
"""Two scaled fp16 planes (H32) against three bf16 planes (P32) and the exact-fp32 MFMA layer kernel: error against the float64
product and time per launch, both networks per launch, on the PPO policy's three hidden-layer shapes at 4096 rows, chained
(each layer's planes feed the next, with the a-priori scales of the bound chain).

    python tools/scratch/split16_probe.py [--rows 4096] [--iters 50]
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


def h32_to_f64(planes, rows, K, inv):
    KC = (K + 31) // 32
    v = planes.view(torch.float16).view(rows, KC, 2, 32).double()
    return ((v[:, :, 0] + v[:, :, 1] / 2048.0).reshape(rows, KC * 32)[:, :K]) * inv.double()[:, None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda")
    L, d, stream = _lib.for_device(dev)
    M = args.rows
    torch.manual_seed(0)
    dims = [388, 1024, 1024, 512]
    nets = 2
    obs = (torch.randn(M, dims[0], device=dev) * 2).clamp(-5, 5).contiguous()
    Ws = [[(torch.randn(dims[l + 1], dims[l], device=dev) / dims[l] ** 0.5).contiguous() for l in range(3)] for _ in range(nets)]
    Bs = [[(torch.randn(dims[l + 1], device=dev) * 0.1).contiguous() for l in range(3)] for _ in range(nets)]
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
    f32 = lambda *s: torch.empty(*s, device=dev)
    # --- weights: H32 planes + per-row scales
    wp, winv = [[None] * 3 for _ in range(nets)], [[None] * 3 for _ in range(nets)]
    for g in range(nets):
        for l in range(3):
            N, K = dims[l + 1], dims[l]
            wp[g][l] = u8(N * ((K + 31) // 32) * 128)
            sc, iv = f32(N), f32(N)
            _lib.check(L.mms_split_planes16_group(d, 1, N, K, 0, ptrs([Ws[g][l]]), ptrs([wp[g][l]]), ptrs([sc]), ptrs([iv]), 0, 0, None, None, None, None, 0.0, stream), what="split16 w", L=L)
            winv[g][l] = iv
            back = h32_to_f64(wp[g][l], N, K, iv)
            rel = float(((back - Ws[g][l].double()).abs() / Ws[g][l].double().abs().clamp_min(1e-30)).max())
            assert rel < 2.0 ** -21, rel
    # --- the chain constants (mult, add) per network and layer
    chain = torch.stack([torch.stack([torch.stack([Ws[g][l].abs().sum(1).max(), Bs[g][l].abs().max()]) for l in range(3)]) for g in range(nets)]).contiguous()
    xp = u8(M * ((dims[0] + 31) // 32) * 128)
    xs, xi = f32(M), f32(M)
    cs, ci = f32(nets, 3, M), f32(nets, 3, M)

    def split_obs():
        _lib.check(L.mms_split_planes16_group(d, 1, M, dims[0], 0, ptrs([obs]), ptrs([xp]), ptrs([xs]), ptrs([xi]), nets, 3, ptrs([chain]), ptrs([cs]), ptrs([ci]), None, 0.0, stream), what="split16 x", L=L)
    split_obs()
    torch.cuda.synchronize()
    back = h32_to_f64(xp, M, dims[0], xi)
    print("obs planes: max rel err %.3e" % float(((back - obs.double()).abs() / obs.double().abs().clamp_min(1e-30)).max()), " scales", xs.min().item(), xs.max().item())
    print("chain scales per layer (min / max):", [(float(cs[0, l].min()), float(cs[0, l].max())) for l in range(3)])
    hp = [[u8(M * (dims[l + 1] // 32) * 128) for l in range(2)] for g in range(nets)]
    y3 = [f32(M, dims[3]) for g in range(nets)]

    def layer16(l):
        xin = [xp] * nets if l == 0 else [hp[g][l - 1] for g in range(nets)]
        xinv = [xi] * nets if l == 0 else [ci[g, l - 1] for g in range(nets)]
        out = [hp[g][l] for g in range(nets)] if l < 2 else y3
        ysc = ptrs([cs[g, l] for g in range(nets)]) if l < 2 else None
        _lib.check(L.mms_linear_group_act_split16(d, nets, M, dims[l + 1], dims[l], ptrs(xin), ptrs([wp[g][l] for g in range(nets)]), ptrs([Bs[g][l] for g in range(nets)]),
                                                  ptrs(out), ptrs(xinv), ptrs([winv[g][l] for g in range(nets)]), ysc, 1, 1 if l < 2 else 0,
                                                  None, None, None, None, None, 0, stream), what="split16 layer", L=L)
    for l in range(3):
        layer16(l)
    torch.cuda.synchronize()
    # error of each layer on the inputs it actually received
    for l in range(3):
        for g in range(nets):
            xin = obs.double() if l == 0 else h32_to_f64(hp[g][l - 1], M, dims[l], ci[g, l - 1])
            ref = torch.nn.functional.elu(xin @ Ws[g][l].double().t() + Bs[g][l].double())
            got = h32_to_f64(hp[g][l], M, dims[l + 1], ci[g, l]) if l < 2 else y3[g].double()
            rms = float(ref.pow(2).mean().sqrt())
            e = got - ref
            # the exact-fp32 kernel on the same (fp32-rounded) inputs
            x32 = xin.float().contiguous()
            y32 = f32(M, dims[l + 1])
            _lib.check(L.mms_linear2_act(d, M, dims[l + 1], dims[l], x32.data_ptr(), Ws[g][l].data_ptr(), Bs[g][l].data_ptr(), y32.data_ptr(), x32.data_ptr(),
                                         Ws[g][l].data_ptr(), Bs[g][l].data_ptr(), y32.data_ptr(), 1, stream), what="linear2", L=L)
            torch.cuda.synchronize()
            ref32 = torch.nn.functional.elu(x32.double() @ Ws[g][l].double().t() + Bs[g][l].double())
            e32 = y32.double() - ref32
            print("layer %d net %d  K %4d N %4d  split16: max %.3e rms %.3e mean %+.2e | exact fp32 mfma: max %.3e rms %.3e mean %+.2e  (units of rms(Y) = %.3f; max|y| %.1f, smallest row scale %g)" %
                  (l, g, dims[l], dims[l + 1], float(e.abs().max()) / rms, float(e.pow(2).mean().sqrt()) / rms, float(e.mean()) / rms,
                   float(e32.abs().max()) / rms, float(e32.pow(2).mean().sqrt()) / rms, float(e32.mean()) / rms, rms, float(ref.abs().max()),
                   float(cs[g, l].min())), flush=True)

    def timeit(fn, name, flops):
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
        print("%-48s %.1f us per launch%s" % (name, us, (" = %.1f TFLOP/s fp32-equivalent" % (flops / us / 1e6)) if flops else ""), flush=True)
    for l in range(3):
        timeit(lambda: layer16(l), "split16 layer %d (K %d N %d, both nets)" % (l, dims[l], dims[l + 1]), 2 * 2.0 * M * dims[l] * dims[l + 1])
    timeit(split_obs, "split16 obs planes + chain scales", 0)

    def all3():
        split_obs()
        for l in range(3):
            layer16(l)
    timeit(all3, "obs split + three layers", 2 * 2.0 * M * sum(dims[l] * dims[l + 1] for l in range(3)))
    # P32 for comparison (same shapes, own planes)
    wp3 = [[u8(dims[l + 1] * ((dims[l] + 31) // 32) * 192) for l in range(3)] for g in range(nets)]
    for g in range(nets):
        for l in range(3):
            _lib.check(L.mms_split_planes(d, dims[l + 1], dims[l], 0, Ws[g][l].data_ptr(), wp3[g][l].data_ptr(), stream), what="split w", L=L)
    xp3 = u8(M * ((dims[0] + 31) // 32) * 192)
    hp3 = [[u8(M * (dims[l + 1] // 32) * 192) for l in range(2)] for g in range(nets)]

    def layer3(l):
        xin = [xp3] * nets if l == 0 else [hp3[g][l - 1] for g in range(nets)]
        out = [hp3[g][l] for g in range(nets)] if l < 2 else y3
        _lib.check(L.mms_linear_group_act_split(d, nets, M, dims[l + 1], dims[l], ptrs(xin), ptrs([wp3[g][l] for g in range(nets)]), ptrs([Bs[g][l] for g in range(nets)]),
                                                ptrs(out), 1, 1 if l < 2 else 0, None, None, None, None, None, 0, stream), what="split layer", L=L)
    _lib.check(L.mms_split_planes(d, M, dims[0], 0, obs.data_ptr(), xp3.data_ptr(), stream), what="split x", L=L)
    for l in range(3):
        layer3(l)
    for l in range(3):
        timeit(lambda: layer3(l), "bf16x3 layer %d (K %d N %d, both nets)" % (l, dims[l], dims[l + 1]), 2 * 2.0 * M * dims[l] * dims[l + 1])


if __name__ == "__main__":
    main()
