#!/usr/bin/env python3
"""Episodes lost in the last third of the training-loop test's run (tests/test_gpu_parity.py::test_ppo_training_loop_on_the_hip_path),
per library build (MMS_LIB) and seed: how much the figure moves with the arithmetic's last bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import train_ppo_demo
base = ["--task", "OneAnt", "--num-envs", "1024", "--iterations", "90", "--hidden", "256", "128", "128", "--split-min-tiles", "0", "--log-every", "1000"]
res = []
for seed in (sys.argv[1:] or ["0"]):
    for planes in (False, True):
        args = train_ppo_demo.parse(base + ["--seed", seed] + (["--obs-planes"] if planes else []))
        out = train_ppo_demo.train(args, log=lambda m: None)
        res.append((seed, planes, sum(c for _, c in out["episodes"][60:]), sum(c for _, c in out["episodes"])))
        out["env"].task.engine.close()
print(os.environ.get("MMS_LIB", "default").split("/")[-1], res)
