#!/usr/bin/env python3
"""Where does the one-off stall of tens of ms in the first timed batch of hipGraph replays come from?  (bench.py's sim-only series
measured [80.9, 3.8, 3.7, 3.7] ms for four equal batches in round 1.)

For every replay of a 16-step sim graph this prints the host time spent INSIDE graph.replay() (enqueue) and the wall time until
the device has finished it -- first with a synchronize after every replay, then in un-synchronised batches like bench.py's.
A stall inside replay() is host-side runtime work (graph upload, kernarg pools, code-object load); a stall between enqueue and
completion is the device (clock ramp, first-touch page faults).  Variants: --same-stream replays on the capture stream,
--idle MS sleeps between batches (a GPU that idles drops its clocks).

    python tools/graph_stall_probe.py [--same-stream] [--idle 50]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--same-stream", action="store_true")
    ap.add_argument("--idle", type=float, default=0.0)
    ap.add_argument("--prereplay", type=int, default=8)
    a = ap.parse_args()
    import torch
    from massive_marl_benchmark_amd.engine import Engine
    N = 4096
    eng = Engine("TenAnt", num_envs=N, device=0, seed=0, clip_obs=5.0)
    g = torch.Generator().manual_seed(1234)
    ring = [(torch.rand(N, 80, generator=g) * 2 - 1).cuda() for _ in range(16)]
    act = eng.tensor("actions")

    def sim_step(i):
        act.copy_(ring[i % 16]); eng.step()
    for i in range(64):
        sim_step(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for i in range(16):
                sim_step(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ctx = torch.cuda.stream(side) if a.same_stream else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        sync = side.synchronize if a.same_stream else torch.cuda.synchronize
        print("# phase 1: %d replays, synchronize after each: [enqueue ms, total ms]" % 24)
        for i in range(24):
            t0 = time.perf_counter()
            graph.replay()
            t1 = time.perf_counter()
            sync()
            t2 = time.perf_counter()
            print("replay %2d  enqueue %8.3f  total %8.3f" % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
        print("# phase 2: batches of 8 replays, one synchronize per batch (bench.py's pattern), idle %.0f ms between" % a.idle)
        for b in range(10):
            if a.idle:
                time.sleep(a.idle * 1e-3)
            enq = []
            t0 = time.perf_counter()
            for _ in range(8):
                s = time.perf_counter()
                graph.replay()
                enq.append((time.perf_counter() - s) * 1e3)
            sync()
            t2 = time.perf_counter()
            print("batch %2d  total %8.3f ms (%.1f us/step)  max enqueue %7.3f  enqueues %s" % (
                b, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / 128, max(enq), " ".join("%.2f" % e for e in enq)), flush=True)
        print("# phase 3: eager, 128 steps per batch")
        for b in range(4):
            t0 = time.perf_counter()
            for i in range(128):
                sim_step(i)
            sync()
            print("eager batch %d  total %8.3f ms" % (b, (time.perf_counter() - t0) * 1e3), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
