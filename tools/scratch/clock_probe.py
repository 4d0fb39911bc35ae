"""Probe: after a cache-hungry kernel, time the FIRST and the SECOND of two consecutive step launches separately."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd.engine import Engine
N = 4096
eng = Engine("TenAnt", num_envs=N, device=0, seed=0)
g = torch.Generator().manual_seed(1)
ring = [(torch.rand(N, 80, generator=g) * 2 - 1).cuda() for _ in range(16)]
act = eng.tensor("actions")
for i in range(200):
    act.copy_(ring[i % 16]); eng.step()
rd = torch.empty(16 * 1024 * 1024, device="cuda")
def nothing(): pass
def read_sweep(): rd.sum()
for name, pre in (("nothing", nothing), ("read sweep 64 MB", read_sweep)):
    t1 = t2 = t3 = 0.0
    n = 48
    for i in range(n + 8):
        act.copy_(ring[i % 16])
        pre()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record(); eng.step(); ev[1].record(); eng.step(); ev[2].record(); eng.step(); ev[3].record()
        torch.cuda.synchronize()
        if i >= 8:
            t1 += ev[0].elapsed_time(ev[1]); t2 += ev[1].elapsed_time(ev[2]); t3 += ev[2].elapsed_time(ev[3])
    print("%-20s first %.1f us  second %.1f us  third %.1f us" % (name, t1 / n * 1e3, t2 / n * 1e3, t3 / n * 1e3), flush=True)
