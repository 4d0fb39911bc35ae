"""Probe: fixed cost of a mms_linear2_act launch (both networks, M = 4096, N = 1024) -- time against K, back to back."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
M, N = 4096, 1024
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K in (32, 64, 128, 256, 512, 1024, 2048):
    x0, x1 = torch.randn(M, K, device="cuda"), torch.randn(M, K, device="cuda")
    w0, w1 = torch.randn(N, K, device="cuda"), torch.randn(N, K, device="cuda")
    b0, b1 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    y0, y1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    def ours():
        assert L.mms_linear2_act(0, M, N, K, p(x0), p(w0), p(b0), p(y0), p(x1), p(w1), p(b1), p(y1), 1, st) == 0
    t = timeit(ours)
    print("K=%5d  %.1f us  (%.2f us per 32-wide slice beyond the first)" % (K, t, 0.0 if K == 32 else (t - base) / (K / 32 - 1)), flush=True) if K != 32 else print("K=%5d  %.1f us" % (K, t), flush=True)
    if K == 32: base = t
y = torch.empty(2, M, N, device="cuda")
print("33.5 MB fill (torch zero_): %.1f us" % timeit(lambda: y.zero_()))
z = torch.empty(1, device="cuda")
print("empty-ish kernel (1-element fill): %.1f us" % timeit(lambda: z.zero_()))
