"""Probe: step-kernel duration (HIP events, back-to-back launches) as a function of the action distribution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd.engine import Engine
N = 4096
for name, gen in (("uniform(-1,1)", lambda g: torch.rand(N, 80, generator=g) * 2 - 1),
                  ("normal(0,0.64)", lambda g: torch.randn(N, 80, generator=g) * 0.64),
                  ("normal(0,0.1)", lambda g: torch.randn(N, 80, generator=g) * 0.1),
                  ("zeros", lambda g: torch.zeros(N, 80))):
    eng = Engine("TenAnt", num_envs=N, device=0, seed=0)
    g = torch.Generator().manual_seed(1)
    ring = [gen(g).cuda() for _ in range(16)]
    act = eng.tensor("actions")
    for i in range(300):
        act.copy_(ring[i % 16]); eng.step()
    torch.cuda.synchronize()
    r0 = int(eng.tensor("reset_count").sum())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(64):
        act.copy_(ring[i % 16])
        e0.record(); eng.step(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    r1 = int(eng.tensor("reset_count").sum())
    z = eng.tensor("root_states").view(N, 11, 13)[:, :10, 2]
    print("%-16s step %.1f us  resets/step %.1f  mean torso z %.3f" % (name, tot / 64 * 1e3, (r1 - r0) / 64.0, float(z.mean())), flush=True)
    eng.close()
