"""One shape of the two-fp16-plane layer kernel launched 20 times (both networks, 4096 x 1024 x 1024, planes out): the target of rocprofv3 --pmc passes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
M, K, N = 4096, 1024, 1024
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
nb = lambda r, k: r * ((k + 31) // 32) * 128
f32 = lambda n: torch.empty(n, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = [torch.randn(M, K, device="cuda") for _ in range(2)]
w = [torch.randn(N, K, device="cuda") / 32 for _ in range(2)]
xp = [torch.empty(nb(M, K), dtype=torch.uint8, device="cuda") for _ in range(2)]
wp = [torch.empty(nb(N, K), dtype=torch.uint8, device="cuda") for _ in range(2)]
xs, xi, ws, wi = [[f32(n) for _ in range(2)] for n in (M, M, N, N)]
L.mms_split_planes16_group(0, 2, M, K, 0, arr(x), arr(xp), arr(xs), arr(xi), 0, 0, None, None, None, None, 0.0, st)
L.mms_split_planes16_group(0, 2, N, K, 0, arr(w), arr(wp), arr(ws), arr(wi), 0, 0, None, None, None, None, 0.0, st)
b = [torch.zeros(N, device="cuda") for _ in range(2)]
ysc = [torch.full((M,), 64.0, device="cuda") for _ in range(2)]
y = [torch.empty(nb(M, N), dtype=torch.uint8, device="cuda") for _ in range(2)]
for _ in range(20):
    assert L.mms_linear_group_act_split16(0, 2, M, N, K, arr(xp), arr(wp), arr(b), arr(y), arr(xi), arr(wi), arr(ysc), 1, 1, None, None, None, None, None, 0, st) == 0
torch.cuda.synchronize()
