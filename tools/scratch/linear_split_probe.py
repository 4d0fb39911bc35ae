"""Probe: one network per launch at M = 4096 (actor per step) and at M = 32768 (critic batched over a rollout of 8 steps)
against both networks per launch at M = 4096 (what the rollout step does now)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = {"dual": 0.0, "single": 0.0, "batched8": 0.0}
for (K, N) in ((388, 1024), (1024, 1024), (1024, 512)):
    res = {}
    for name, M, two in (("dual", 4096, True), ("single", 4096, False), ("batched8", 32768, False)):
        x0, x1 = torch.randn(M, K, device="cuda"), torch.randn(M, K, device="cuda")
        w0, w1 = torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, K, device="cuda") / K ** 0.5
        b0, b1 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        y0, y1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        if two:
            f = lambda: L.mms_linear2_act(0, M, N, K, p(x0), p(w0), p(b0), p(y0), p(x1), p(w1), p(b1), p(y1), 1, st)
        else:
            f = lambda: L.mms_linear2_act(0, M, N, K, p(x0), p(w0), p(b0), p(y0), None, None, None, None, 1, st)
        res[name] = timeit(f)
        tot[name] += res[name]
        del x0, x1, y0, y1
    print("K=%4d N=%4d  dual %.1f us   single %.1f us   batched x8 %.1f us (%.1f per step, %.0f TF)" % (K, N, res["dual"], res["single"], res["batched8"], res["batched8"] / 8, 2 * 32768 * K * N / res["batched8"] / 1e6), flush=True)
print("per rollout step: now %.1f us; actor per step + critic batched %.1f us" % (tot["dual"], tot["single"] + tot["batched8"] / 8))
