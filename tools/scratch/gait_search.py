"""Physics sanity probe: can the ant of this build's model walk?  4096 OneAnt envs, each driven open loop by its own random
sinusoidal gait (per-joint amplitude, phase, offset; common frequency); reports the distance covered by the survivors."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd.engine import Engine
N, STEPS = 4096, 360                       # 6 s
eng = Engine("OneAnt", num_envs=N, device=0, seed=0)
g = torch.Generator().manual_seed(0)
amp = torch.rand(N, 8, generator=g).cuda()
phase = (torch.rand(N, 8, generator=g) * 2 * math.pi).cuda()
offs = ((torch.rand(N, 8, generator=g) - 0.5) * 0.6).cuda()
freq = (1.0 + 2.5 * torch.rand(N, 1, generator=g)).cuda()
act = eng.tensor("actions")
act.zero_(); eng.step()                    # reset step
root = eng.tensor("root_states").view(N, 2, 13)
x0 = root[:, 0, 0:2].clone()
alive = torch.ones(N, dtype=torch.bool, device="cuda")
dt = 0.0166
for t in range(STEPS):
    act.copy_(torch.clamp(offs + amp * torch.sin(2 * math.pi * freq * (t * dt) + phase), -1, 1))
    eng.step()
    alive &= eng.tensor("reset") == 0
    alive &= eng.tensor("reset_count") == 1
torch.cuda.synchronize()
d = (root[:, 0, 0:2] - x0).norm(dim=-1)
dx = (root[:, 0, 0] - x0[:, 0])
print("survivors %d / %d" % (int(alive.sum()), N))
ds = d[alive]
print("distance covered in %.1f s by survivors: median %.2f m, 90th pct %.2f m, max %.2f m (%.2f m/s)" % (STEPS * dt, float(ds.median()), float(ds.quantile(0.9)), float(ds.max()), float(ds.max()) / (STEPS * dt)))
print("max forward (+x, towards the box) %.2f m; torso height of the best: %.2f m; max speed seen %.1f m/s" % (float(dx[alive].max()), float(root[alive][ds.argmax(), 0, 2]), float(root[:, 0, 7:10].norm(dim=-1).max())))
