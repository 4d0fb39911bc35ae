"""Probe: mms_linear2_act (both networks, bias + ELU fused) against torch Linear + ELU on two streams / one stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda"
M = 4096
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (K, N) in ((388, 1024), (1024, 1024), (1024, 512)):
    x0, x1 = torch.randn(M, K, device=dev), torch.randn(M, K, device=dev)
    w0, w1 = torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, K, device=dev) / K ** 0.5
    b0, b1 = torch.randn(N, device=dev), torch.randn(N, device=dev)
    y0, y1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    def ours():
        rc = L.mms_linear2_act(0, M, N, K, p(x0), p(w0), p(b0), p(y0), p(x1), p(w1), p(b1), p(y1), 1, st)
        assert rc == 0, _lib.last_error(None)
    def ref():
        return torch.nn.functional.elu(torch.nn.functional.linear(x0, w0, b0)), torch.nn.functional.elu(torch.nn.functional.linear(x1, w1, b1))
    ours(); torch.cuda.synchronize()
    r0, r1 = ref()
    scale = (x0.abs() @ w0.abs().t() + b0.abs())
    err = ((y0 - r0).abs() / scale).max().item(), ((y1 - r1).abs() / (x1.abs() @ w1.abs().t() + b1.abs())).max().item()
    t_ours, t_ref = timeit(ours), timeit(ref)
    fl = 2 * 2 * M * K * N / 1e6
    print("K=%4d N=%4d  ours %.1f us (%.0f TF)  torch linear+elu x2 %.1f us (%.0f TF)  max rel err %.1e %.1e" % (K, N, t_ours, fl / t_ours, t_ref, fl / t_ref, err[0], err[1]), flush=True)
