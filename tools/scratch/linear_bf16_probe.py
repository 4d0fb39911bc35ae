"""Probe: mms_linear2_act_bf16 (both networks, bias + ELU fused, bf16 out) against torch bf16 Linear + ELU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
dev = "cuda"
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
bf = torch.bfloat16
for (K, N, xf32) in ((388, 1024, True), (1024, 1024, False), (1024, 512, False)):
    ldw = (K + 63) // 64 * 64
    xs = [torch.randn(M, K, device=dev) for _ in range(2)]
    if not xf32:
        xs = [x.to(bf) for x in xs]
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(bf) for _ in range(2)]
    wp = []
    for w in ws:
        q = torch.zeros(N, ldw, device=dev, dtype=bf); q[:, :K] = w; wp.append(q)
    bs = [torch.randn(N, device=dev) for _ in range(2)]
    ys = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(2)]
    def ours():
        rc = L.mms_linear2_act_bf16(0, M, N, K, ldw, 1 if xf32 else 0, p(xs[0]), p(wp[0]), p(bs[0]), p(ys[0]), p(xs[1]), p(wp[1]), p(bs[1]), p(ys[1]), 1, st)
        assert rc == 0, _lib.last_error(None)
    bb = [b.to(bf) for b in bs]
    def ref():
        return [torch.nn.functional.elu(torch.nn.functional.linear(x.to(bf), w, b)) for x, w, b in zip(xs, ws, bb)]
    ours(); torch.cuda.synchronize()
    exact = [torch.nn.functional.elu(torch.nn.functional.linear(x.to(bf).double(), w.double(), b.double())) for x, w, b in zip(xs, ws, bs)]
    err = [float(((y.double() - e).abs() / (1.0 + e.abs())).max()) for y, e in zip(ys, exact)]
    err_ref = [float(((y.double() - e).abs() / (1.0 + e.abs())).max()) for y, e in zip(ref(), exact)]
    t_ours, t_ref = timeit(ours), timeit(ref)
    fl = 2 * 2 * M * K * N / 1e6
    print("K=%4d N=%4d  ours %.1f us (%.0f TF)  torch bf16 linear+elu x2 %.1f us (%.0f TF)  max err vs fp64: ours %.1e torch %.1e" % (K, N, t_ours, fl / t_ours, t_ref, fl / t_ref, max(err), max(err_ref)), flush=True)
