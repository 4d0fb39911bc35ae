import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from massive_marl_benchmark_amd import _lib
L, C = _lib.lib(), _lib.lib_cpu()
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(2)
rows, K = 4096, 388
x = torch.randn(rows, K, device="cuda") * torch.exp2(torch.randint(-24, 16, (rows, 1), device="cuda").float())
x[0, :8] = torch.tensor([0.0, -0.0, 1e-30, -3e4, 1.0 + 2 ** -23, 2 ** -20, 65504.0, -1e-20], device="cuda")
nb = rows * ((K + 31) // 32) * 128
planes = torch.zeros(nb, dtype=torch.uint8, device="cuda"); sc = torch.empty(rows, device="cuda"); iv = torch.empty(rows, device="cuda")
L.mms_split_planes16_group(0, 1, rows, K, 0, arr([x]), arr([planes]), arr([sc]), arr([iv]), 0, 0, None, None, None, None, 0.0, stream)
torch.cuda.synchronize()
xc = x.cpu(); pc = torch.zeros(nb, dtype=torch.uint8); scc = torch.empty(rows); ivc = torch.empty(rows)
C.mms_split_planes16_group(-1, 1, rows, K, 0, arr([xc]), arr([pc]), arr([scc]), arr([ivc]), 0, 0, None, None, None, None, 0.0, None)
print("scales equal", torch.equal(sc.cpu(), scc))
g = planes.cpu().view(torch.int16).view(rows, -1, 2, 32); c = pc.view(torch.int16).view(rows, -1, 2, 32)
d = (g != c)
print("differing halves:", int(d.sum()), "of", d.numel(), " hi:", int(d[:, :, 0].sum()), " lo:", int(d[:, :, 1].sum()))
idx = d.nonzero()[:12]
for r, kc, pl, j in idx.tolist():
    k = kc * 32 + j
    xv = float(xc[r, k]) if k < K else 0.0
    print("row %d k %d plane %d  x %.9g scaled %.9g  gpu %s (%.6g) cpu %s (%.6g)  hi gpu %.6g" % (r, k, pl, xv, xv * float(scc[r]), hex(g[r, kc, pl, j].item() & 0xffff),
          float(g[r, kc, pl, j].view(torch.float16)), hex(c[r, kc, pl, j].item() & 0xffff), float(c[r, kc, pl, j].view(torch.float16)), float(g[r, kc, 0, j].view(torch.float16))))
