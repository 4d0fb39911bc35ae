"""Probe: the two networks of a layer as ONE launch (both per launch) against TWO single-network launches on two streams,
issued together and with the second delayed by a spin kernel of d us (staggered phases)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
M, K, N = 4096, 1024, 1024
x0, x1 = torch.randn(M, K, device="cuda"), torch.randn(M, K, device="cuda")
w0, w1 = torch.randn(N, K, device="cuda") / 32, torch.randn(N, K, device="cuda") / 32
b0, b1 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
y0, y1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sp = lambda s: ctypes.c_void_p(s.cuda_stream)
def dual():
    L.mms_linear2_act(0, M, N, K, p(x0), p(w0), p(b0), p(y0), p(x1), p(w1), p(b1), p(y1), 1, sp(s1))
pad = torch.empty(1 << 20, device="cuda")
def two(delay_elems):
    L.mms_linear2_act(0, M, N, K, p(x0), p(w0), p(b0), p(y0), None, None, None, None, 1, sp(s1))
    with torch.cuda.stream(s2):
        if delay_elems:
            pad[:delay_elems].add_(1.0)          # a small kernel in front of the second launch: shifts its start
    L.mms_linear2_act(0, M, N, K, p(x1), p(w1), p(b1), p(y1), None, None, None, None, 1, sp(s2))
def timeit(f, n=40, layers=3):
    # `layers` dependent repetitions per sample, like the chain of a policy; streams joined at the end of each sample
    def sample():
        for _ in range(layers): f()
        s1.wait_stream(s2)
        s2.wait_stream(s1)
    for _ in range(3): sample()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    for _ in range(n): sample()
    e1.record(s1); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * layers) * 1e3
print("both networks per launch:          %.1f us per layer" % timeit(dual))
for d in (0, 1 << 12, 1 << 16, 1 << 19, 1 << 20):
    print("two launches, two streams, pad %7d elems: %.1f us per layer" % (d, timeit(lambda: two(d))), flush=True)
