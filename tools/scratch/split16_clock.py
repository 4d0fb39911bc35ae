#!/usr/bin/env python3
"""Clock held by block 0 during one layer launch (builds with -DMMS_S16_STAMP=1 only): shader cycles / 100-MHz ticks."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from massive_marl_benchmark_amd import _lib  # noqa: E402
L, d, stream = _lib.for_device(torch.device("cuda"))
M, N, K = 4096, 1024, int(os.environ.get("PROBE_K", "1024"))
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
nb = lambda r, K: r * ((K + 31) // 32) * 128
f32 = lambda n: torch.empty(n, device="cuda")
x = [torch.randn(M, K, device="cuda") for _ in range(2)]
w = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(2)]
xp = [torch.empty(nb(M, K), dtype=torch.uint8, device="cuda") for _ in range(2)]
wp = [torch.empty(nb(N, K), dtype=torch.uint8, device="cuda") for _ in range(2)]
xs, xi, ws, wi = [[f32(n) for _ in range(2)] for n in (M, M, N, N)]
L.mms_split_planes16_group(d, 2, M, K, 0, arr(x), arr(xp), arr(xs), arr(xi), 0, 0, None, None, None, None, 0.0, stream)
L.mms_split_planes16_group(d, 2, N, K, 0, arr(w), arr(wp), arr(ws), arr(wi), 0, 0, None, None, None, None, 0.0, stream)
b = [torch.zeros(N, device="cuda") for _ in range(2)]
ysc = [torch.full((M,), 64.0, device="cuda") for _ in range(2)]
y = [torch.empty(nb(M, N), dtype=torch.uint8, device="cuda") for _ in range(2)]
fn = lambda: L.mms_linear_group_act_split16(d, 2, M, N, K, arr(xp), arr(wp), arr(b), arr(y), arr(xi), arr(wi), arr(ysc), 1, 1, None, None, None, None, None, 0, stream)
out = []
for i in range(60):
    assert fn() == 0
    if i >= 50:
        torch.cuda.synchronize()
        c, r, r1, r2 = y[0][:32].view(torch.int64).tolist()
        out.append("%.0f cyc / %.2f us = %.2f GHz; first slice landed at %.2f us, k-loop done at %.2f us, block done at %.2f us" % (c, r / 100.0, c / (r * 10.0), r1 / 100.0, r2 / 100.0, r / 100.0))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    fn()
e1.record()
torch.cuda.synchronize()
print(os.environ.get("MMS_LIB", "default").split("/")[-1], "K", K, "| launch to launch %.2f us |" % (e0.elapsed_time(e1) * 10), " || ".join(out[-2:]))
