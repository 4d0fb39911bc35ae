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
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    L.mms_linear2_act(0, M, N, K, p(x0), p(w0), p(b0), p(y0), p(x1), p(w1), p(b1), p(y1), 1, st)
    torch.nn.functional.linear(x0, w0, b0)
torch.cuda.synchronize()
