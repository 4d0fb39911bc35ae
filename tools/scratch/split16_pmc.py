"""One shape of the two-fp16-plane layer kernel launched 20 times (default: both networks, 4096 x 1024 x 1024, planes out; PMC_G / PMC_M / PMC_K /
PMC_N in the environment: another shape, e.g. the grouped MARL layers 20 x 4096 x 512 x 512): the target of rocprofv3 --pmc passes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
G = int(os.environ.get("PMC_G", 2))
M, K, N = int(os.environ.get("PMC_M", 4096)), int(os.environ.get("PMC_K", 1024)), int(os.environ.get("PMC_N", 1024))
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
nb = lambda r, k: r * ((k + 31) // 32) * 128
f32 = lambda n: torch.empty(n, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = [torch.randn(M, K, device="cuda") for _ in range(G)]
w = [torch.randn(N, K, device="cuda") / 32 for _ in range(G)]
xp = [torch.empty(nb(M, K), dtype=torch.uint8, device="cuda") for _ in range(G)]
wp = [torch.empty(nb(N, K), dtype=torch.uint8, device="cuda") for _ in range(G)]
xs, xi, ws, wi = [[f32(n) for _ in range(G)] for n in (M, M, N, N)]
L.mms_split_planes16_group(0, G, M, K, 0, arr(x), arr(xp), arr(xs), arr(xi), 0, 0, None, None, None, None, 0.0, st)
L.mms_split_planes16_group(0, G, N, K, 0, arr(w), arr(wp), arr(ws), arr(wi), 0, 0, None, None, None, None, 0.0, st)
b = [torch.zeros(N, device="cuda") for _ in range(G)]
ysc = [torch.full((M,), 64.0, device="cuda") for _ in range(G)]
y = [torch.empty(nb(M, N), dtype=torch.uint8, device="cuda") for _ in range(G)]
for _ in range(20):
    assert L.mms_linear_group_act_split16(0, G, M, N, K, arr(xp), arr(wp), arr(b), arr(y), arr(xi), arr(wi), arr(ysc), 1, 1, None, None, None, None, None, 0, st) == 0
torch.cuda.synchronize()
