"""Probe: mms_linear2_act with ONE network (the critic's bootstrap pass) against torch Linear + ELU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
M = 4096
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for env in ("", "MMS_LINEAR_SMALL_MAX=300"):
    for (K, N) in ((388, 1024), (1024, 1024), (1024, 512)):
        x, w, b, y = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda"), torch.empty(M, N, device="cuda")
        def ours():
            assert L.mms_linear2_act(0, M, N, K, p(x), p(w), p(b), p(y), None, None, None, None, 1, st) == 0
        def ref():
            return torch.nn.functional.elu(torch.nn.functional.linear(x, w, b))
        print(env, "K=%4d N=%4d  ours (one network) %.1f us   torch linear+elu %.1f us" % (K, N, timeit(ours), timeit(ref)), flush=True)
    break
w1 = torch.randn(1, 512, device="cuda"); b1 = torch.randn(1, device="cuda"); h = torch.randn(M, 512, device="cuda")
print("value head 512 -> 1: F.linear %.1f us, (h * w).sum(-1) %.1f us, torch.mv %.1f us" % (
    timeit(lambda: torch.nn.functional.linear(h, w1, b1)), timeit(lambda: (h * w1).sum(-1) + b1), timeit(lambda: torch.mv(h, w1[0]) + b1)))
