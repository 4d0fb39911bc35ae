#!/usr/bin/env python3
"""HBM traffic of the step kernel from rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
SEPARATE --pmc passes (they do not fit one pass), FETCH_SIZE doubled on gfx950 (it tallies the 128-B requests of wide coalesced
reads at 64 B), WRITE_SIZE as read.  Runs tools/profile_step.py under the profiler (program directly after `--`, no wrappers),
averages per dispatch of the dominant step kernel and writes gpurun_out/step_kernel_traffic.json -- with the hash of the kernel's
sources (bench.step_kernel_source_hash) so that bench.py only quotes the figure for the build it was measured on.  Copy the file
to profiles/step_kernel_traffic.json.  Also collects SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_WAVES in a third pass.

    python tools/pmc_traffic.py            (on the GPU box)
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")


def run_pass(tag, counters, extra):
    d = os.path.join(OUT, "pmc_" + tag)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", sys.executable,
                                                os.path.join(ROOT, "tools", "profile_step.py"), "--steps", "128"] + extra
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    step = [k for k in acc if "ant_step_kernel" in k]
    assert step, "no step kernel in the counter output: %s" % list(acc)[:5]
    k = max(step, key=lambda n: len(next(iter(acc[n].values()))))
    return k, {c: (sum(v) / len(v), len(v)) for c, v in acc[k].items()}


def main():
    from bench import ALGO_BYTES_PER_ENV_STEP, step_kernel_source_hash
    os.makedirs(OUT, exist_ok=True)
    res = {}
    for label, extra in (("rollout", ["--rollout-outputs"]), ("engine_buffers", [])):
        kf, f = run_pass(label + "_fetch", ["FETCH_SIZE"], extra)
        kw, w = run_pass(label + "_write", ["WRITE_SIZE"], extra)
        fetch_kb, n = f["FETCH_SIZE"]
        write_kb, _ = w["WRITE_SIZE"]
        res[label] = {"kernel": kf, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "dispatches": n,
                      "traffic_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024))}
    ki, inst = run_pass("insts", ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVES"], ["--rollout-outputs"])
    waves = inst["SQ_WAVES"][0]
    out = {
        "kernel": res["rollout"]["kernel"], "num_envs": 4096, "source_hash": step_kernel_source_hash(),
        "configuration": "rollout outputs: one clamped observation row per env-step written into the bound rollout slot + reward / done "
                         "slots (what bench.py's rollout step launches)",
        "fetch_size_kb": res["rollout"]["fetch_size_kb"], "write_size_kb": res["rollout"]["write_size_kb"], "fetch_correction": 2.0,
        "traffic_bytes_per_launch": res["rollout"]["traffic_bytes_per_launch"],
        "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * 4096,
        "source": "tools/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/profile_step.py --steps 128 "
                  "--rollout-outputs, %d dispatches averaged; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 tallies wide coalesced reads "
                  "at half their bytes)" % res["rollout"]["dispatches"],
        "engine_buffers_configuration": dict(res["engine_buffers"], what="stand-alone engine defaults: raw + clamped observation buffers both "
                                                                         "written (sim-only series)"),
        "instructions": {"SQ_INSTS_VALU": inst["SQ_INSTS_VALU"][0], "SQ_INSTS_LDS": inst["SQ_INSTS_LDS"][0], "SQ_WAVES": waves,
                         "valu_per_wave": inst["SQ_INSTS_VALU"][0] / waves, "lds_per_wave": inst["SQ_INSTS_LDS"][0] / waves},
    }
    path = os.path.join(OUT, "step_kernel_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
