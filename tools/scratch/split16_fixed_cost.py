#!/usr/bin/env python3
"""Time of the two-fp16-plane layer kernel against K (both networks, 4096 rows): the slope is the cost of a 32-wide k-step, the
intercept what a launch pays besides; K values whose row pitch (K / 32 x 128 B) is not a power of two ride along (an L2-channel
camping check: all rows of a k-slice are one pitch apart)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from massive_marl_benchmark_amd import _lib  # noqa: E402

L, d, stream = _lib.for_device(torch.device("cuda"))
M = 4096
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
nb = lambda r, K: r * ((K + 31) // 32) * 128
f32 = lambda n: torch.empty(n, device="cuda")
for N, planes in ((1024, 1), (512, 0), (1024, 0)):
    res = []
    for K in (32, 64, 128, 256, 512, 992, 1024, 1056, 2016, 2048):
        x = [torch.randn(M, K, device="cuda") for _ in range(2)]
        w = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(2)]
        xp = [torch.empty(nb(M, K), dtype=torch.uint8, device="cuda") for _ in range(2)]
        wp = [torch.empty(nb(N, K), dtype=torch.uint8, device="cuda") for _ in range(2)]
        xs, xi, ws, wi = [[f32(n) for _ in range(2)] for n in (M, M, N, N)]
        L.mms_split_planes16_group(d, 2, M, K, 0, arr(x), arr(xp), arr(xs), arr(xi), 0, 0, None, None, None, None, 0.0, stream)
        L.mms_split_planes16_group(d, 2, N, K, 0, arr(w), arr(wp), arr(ws), arr(wi), 0, 0, None, None, None, None, 0.0, stream)
        b = [torch.zeros(N, device="cuda") for _ in range(2)]
        ysc = [torch.full((M,), 64.0, device="cuda") for _ in range(2)]
        y = [torch.empty(nb(M, N) if planes else M * N * 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
        px, pw, pb, py, pxi, pwi, pys = arr(xp), arr(wp), arr(b), arr(y), arr(xi), arr(wi), arr(ysc)
        fn = lambda: L.mms_linear_group_act_split16(d, 2, M, N, K, px, pw, pb, py, pxi, pwi, pys if planes else None, 1, planes, None, None, None, None, None, 0, stream)
        for _ in range(10):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append((K, e0.elapsed_time(e1) * 10))
    line = "  ".join("K %d: %.1f us" % r for r in res)
    t = dict(res)
    slope = (t[2048] - t[512]) / ((2048 - 512) / 32)
    print("N %4d planes_out %d | %s | per k-step %.2f us, intercept %.1f us" % (N, planes, line, slope, t[2048] - slope * 64), flush=True)
