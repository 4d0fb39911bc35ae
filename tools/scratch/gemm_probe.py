"""Probe: fp32 GEMM times of the PPO policy shapes with / without TunableOp, and with padded K."""
import os, sys, time
import torch
tun = len(sys.argv) > 1 and sys.argv[1] == "tune"
if tun:
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(30)
    torch.cuda.tunable.set_filename("/tmp/tunable.csv")
dev = torch.device("cuda:0")
N = 4096
shapes = [(388, 1024), (392, 1024), (448, 1024), (1024, 1024), (1024, 512), (512, 80), (512, 1), (512, 128)]
for (k, n) in shapes:
    lin = torch.nn.Linear(k, n).to(dev)
    x = torch.randn(N, k, device=dev)
    with torch.no_grad():
        for _ in range(5):
            y = lin(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = lin(x)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("%s K=%4d N=%4d  %.1f us  %.1f TFLOP/s" % ("tuned" if tun else "plain", k, n, us, 2 * N * k * n / us / 1e6), flush=True)
