import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd import _lib
L = _lib.lib()
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
N, H, A = 4096, 512, 80
dev = "cuda"
hid = torch.randn(N, H, device=dev); W = torch.randn(A, H, device=dev) / 22; b = torch.randn(A, device=dev)
ls = torch.zeros(A, device=dev); val = torch.randn(N, device=dev)
cnt = torch.zeros(N, dtype=torch.int64, device=dev)
act, mu, sg = (torch.zeros(N, A, device=dev) for _ in range(3)); logp = torch.zeros(N, device=dev); vs = torch.zeros(N, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def head():
    L.mms_ppo_head_act(0, p(hid), p(W), p(b), H, p(val), p(ls), 1, p(cnt), 0, 1, p(act), p(act), p(logp), p(vs), p(mu), p(sg), N, A, st)
def split():
    m = torch.nn.functional.linear(hid, W, b)
    L.mms_ppo_act(0, p(m), p(val), p(ls), 1, p(cnt), 0, 1, p(act), p(act), p(logp), p(vs), p(mu), p(sg), N, A, st)
def lin():
    torch.nn.functional.linear(hid, W, b)
for name, f in (("head", head), ("linear+act", split), ("linear", lin)):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us" % (e0.elapsed_time(e1) * 10))
