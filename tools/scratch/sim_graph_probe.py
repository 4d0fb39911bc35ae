"""Probe: wall time of successive replays of a 16-step sim graph (default stream vs side stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from massive_marl_benchmark_amd.engine import Engine
N = 4096
eng = Engine("TenAnt", num_envs=N, device=0, seed=0, clip_obs=5.0)
if len(sys.argv) > 1 and sys.argv[1] == "policy":
    from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
    from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage
    ac = ActorCritic((388,), (0,), (80,), 0.8, {"pi_hid_sizes": [1024, 1024, 512], "vf_hid_sizes": [1024, 1024, 512], "activation": "elu"}, seed=1234).cuda()
    storage = RolloutStorage(N, 8, (388,), (0,), (80,), device="cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "setdev":
    torch.cuda.set_device(torch.device("cuda", 0))
g = torch.Generator().manual_seed(1234)
ring = [(torch.rand(N, 80, generator=g) * 2 - 1).cuda() for _ in range(16)]
act = eng.tensor("actions")
def sim_step(i):
    act.copy_(ring[i % 16]); eng.step()
for i in range(64): sim_step(i)
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(16): sim_step(i)
torch.cuda.current_stream().wait_stream(side)
graph.replay(); torch.cuda.synchronize()
for rep in range(6):
    t0 = time.perf_counter()
    for _ in range(32): graph.replay()
    torch.cuda.synchronize()
    print("default stream: 512 steps in %.2f ms (%.1f us per step)" % ((time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e6 / 512), flush=True)
with torch.cuda.stream(side):
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(32): graph.replay()
        side.synchronize()
        print("side stream:    512 steps in %.2f ms (%.1f us per step)" % ((time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e6 / 512), flush=True)
t0 = time.perf_counter()
for i in range(512): sim_step(i)
torch.cuda.synchronize()
print("eager:          512 steps in %.2f ms" % ((time.perf_counter() - t0) * 1e3))
