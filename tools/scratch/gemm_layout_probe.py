"""Probe: fp32 GEMM time of the policy shapes, torch Linear layout (x @ W^T, W [N,K]) against a pre-transposed weight (x @ Wt, Wt [K,N])."""
import torch
dev = torch.device("cuda:0")
M = 4096
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (k, n) in [(388, 1024), (1024, 1024), (1024, 512), (512, 80)]:
    x = torch.randn(M, k, device=dev); W = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    Wt = W.t().contiguous()
    out = torch.empty(M, n, device=dev)
    t_lin = timeit(lambda: torch.nn.functional.linear(x, W, b))
    t_nn = timeit(lambda: torch.addmm(b, x, Wt))
    t_out = timeit(lambda: torch.addmm(b, x, Wt, out=out))
    fl = 2 * M * k * n / 1e6
    print("K=%4d N=%4d  linear %.1f us (%.0f TF)   addmm(x, Wt) %.1f us (%.0f TF)   out= %.1f us" % (k, n, t_lin, fl / t_lin, t_nn, fl / t_nn, t_out), flush=True)
# both nets as one batched GEMM
for (k, n) in [(1024, 1024), (1024, 512)]:
    x2 = torch.randn(2, M, k, device=dev); W2 = torch.randn(2, k, n, device=dev)
    t = timeit(lambda: torch.bmm(x2, W2))
    print("bmm 2 x [%d,%d]x[%d,%d]: %.1f us (%.0f TF)" % (M, k, k, n, t, 2 * 2 * M * k * n / 1e6 / t), flush=True)
