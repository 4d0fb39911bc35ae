#!/usr/bin/env python3
"""Time of the two-fp16-plane layer kernel against K for the grouped MARL shape (20 networks x 4096 rows x 512 columns: five
256 x 128 tiles per CU, persistent loop) and for fewer networks: slope = a k-step, intercept / tiles per CU = what a tile pays besides."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from massive_marl_benchmark_amd import _lib  # noqa: E402

L, d, stream = _lib.for_device(torch.device("cuda"))
M, N = 4096, 512
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
nb = lambda r, K: r * ((K + 31) // 32) * 128
f32 = lambda n: torch.empty(n, device="cuda")
for G, planes in ((20, 1), (20, 0), (4, 1), (8, 1)):
    res = []
    for K in (32, 64, 128, 256, 512):
        x = [torch.randn(M, K, device="cuda") for _ in range(G)]
        w = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(G)]
        xp = [torch.empty(nb(M, K), dtype=torch.uint8, device="cuda") for _ in range(G)]
        wp = [torch.empty(nb(N, K), dtype=torch.uint8, device="cuda") for _ in range(G)]
        xs, xi, ws, wi = [[f32(n) for _ in range(G)] for n in (M, M, N, N)]
        L.mms_split_planes16_group(d, G, M, K, 0, arr(x), arr(xp), arr(xs), arr(xi), 0, 0, None, None, None, None, 0.0, stream)
        L.mms_split_planes16_group(d, G, N, K, 0, arr(w), arr(wp), arr(ws), arr(wi), 0, 0, None, None, None, None, 0.0, stream)
        b = [torch.zeros(N, device="cuda") for _ in range(G)]
        ysc = [torch.full((M,), 64.0, device="cuda") for _ in range(G)]
        y = [torch.empty(nb(M, N) if planes else M * N * 4, dtype=torch.uint8, device="cuda") for _ in range(G)]
        px, pw, pb, py, pxi, pwi, pys = arr(xp), arr(wp), arr(b), arr(y), arr(xi), arr(wi), arr(ysc)
        fn = lambda: L.mms_linear_group_act_split16(d, G, M, N, K, px, pw, pb, py, pxi, pwi, pys if planes else None, 1, planes, None, None, None, None, None, 0, stream)
        for _ in range(5):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append((K, e0.elapsed_time(e1) * 25))
    t = dict(res)
    tiles = G * (M // 256) * (N // 128) / 256.0
    slope = (t[512] - t[128]) / 12
    print("G %2d planes_out %d (%.2f tiles per CU) | %s | per k-step of the launch %.2f us = %.2f us per tile-step, intercept %.1f us = %.1f us per tile" %
          (G, planes, tiles, "  ".join("K %d: %.1f us" % r for r in res), slope, slope / max(tiles, 1), t[512] - 16 * slope, (t[512] - 16 * slope) / max(tiles, 1)), flush=True)
