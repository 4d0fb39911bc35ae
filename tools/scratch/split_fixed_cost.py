#!/usr/bin/env python3
"""Time of the split layer kernel against K (both networks, 4096 rows): the slope is the cost of a 32-wide k-step, the intercept
what a launch pays besides (first loads, epilogue: activation + split + stores)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from massive_marl_benchmark_amd import _lib  # noqa: E402

L, d, stream = _lib.for_device(torch.device("cuda"))
M = 4096
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
nb = lambda r, K: r * ((K + 31) // 32) * 192
for N, planes in ((1024, 1), (512, 0), (1024, 0)):
    res = []
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        xp = [torch.randint(0, 255, (nb(M, K),), dtype=torch.uint8, device="cuda") for _ in range(2)]
        x = [torch.randn(M, K, device="cuda") for _ in range(2)]
        w = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(2)]
        wp = [torch.empty(nb(N, K), dtype=torch.uint8, device="cuda") for _ in range(2)]
        for g in range(2):
            L.mms_split_planes(d, M, K, 0, x[g].data_ptr(), xp[g].data_ptr(), stream)
            L.mms_split_planes(d, N, K, 0, w[g].data_ptr(), wp[g].data_ptr(), stream)
        b = [torch.zeros(N, device="cuda") for _ in range(2)]
        y = [torch.empty(nb(M, N) if planes else M * N * 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
        px, pw, pb, py = arr(xp), arr(wp), arr(b), arr(y)
        fn = lambda: L.mms_linear_group_act_split(d, 2, M, N, K, px, pw, pb, py, 1, planes, None, None, None, None, None, 0, stream)
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append((K, e0.elapsed_time(e1) * 10))
    line = "  ".join("K %d: %.1f us" % r for r in res)
    (k1, t1), (k2, t2) = res[-3], res[-1]
    slope = (t2 - t1) / ((k2 - k1) / 32)
    print("N %4d planes_out %d | %s | per k-step %.2f us, intercept %.1f us" % (N, planes, line, slope, t2 - slope * k2 / 32), flush=True)
