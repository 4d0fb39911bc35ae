"""One-off: 400 000 control steps (1.8 simulated hours) of 4096 TenAnt envs under full-range random actions (graph of 10 steps),
checked every 20 000 steps: finite, speeds / heights / joint angles physical, box on the ground, resets keep happening."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd.engine import Engine
N = 4096
eng = Engine("TenAnt", num_envs=N, device=0, seed=11)
g = torch.Generator().manual_seed(17)
ring = [(torch.rand(N, 80, generator=g) * 2 - 1).cuda() for _ in range(10)]
act = eng.tensor("actions")
for i in range(10):
    act.copy_(ring[i]); eng.step()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        for i in range(10):
            act.copy_(ring[i]); eng.step()
torch.cuda.current_stream().wait_stream(side)
t0 = time.time()
worst_v = worst_w = 0.0
for chunk in range(20):
    for _ in range(2000):
        graph.replay()
    torch.cuda.synchronize()
    r = eng.tensor("root_states").view(N, 11, 13)
    ok = bool(torch.isfinite(r).all()) and bool(torch.isfinite(eng.tensor("dof_state")).all()) and bool(torch.isfinite(eng.tensor("obs")).all())
    worst_v = max(worst_v, float(r[:, :, 7:10].abs().max())); worst_w = max(worst_w, float(r[:, :, 10:13].norm(dim=-1).max()))
    q = eng.tensor("dof_state").view(N, 80, 2)[:, :, 0]
    rc = eng.tensor("reset_count")
    print("steps %7d  finite %s  max|v| %.2f  max|w| %.1f  z in [%.3f, %.3f]  box z in [%.3f, %.3f]  max|q| %.2f  resets/env min %d max %d  %.0f s"
          % ((chunk + 1) * 20000, ok, worst_v, worst_w, float(r[:, :10, 2].min()), float(r[:, :10, 2].max()), float(r[:, 10, 2].min()),
             float(r[:, 10, 2].max()), float(q.abs().max()), int(rc.min()), int(rc.max()), time.time() - t0), flush=True)
    assert ok and float(r[:, :10, 2].max()) < 3.0 and float(r[:, :10, 2].min()) > 0.0 and float(q.abs().max()) < 2.0
