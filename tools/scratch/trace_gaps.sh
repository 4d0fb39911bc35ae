#!/bin/bash
# Kernel timeline of the bench's rollout graph: name, duration and the gap to the previous kernel's end, for one rollout of 8 steps
# taken from the end of the timed region.  Output: gpurun_out/trace_gaps.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_gaps -- python3 $R/bench.py --steps 64 --warmup 16 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = sorted(glob.glob(R + "/gpurun_out/trace_gaps/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last 3 occurrences of the GAE kernel and print the span between the last two
idx = [i for i, r in enumerate(rows) if "gae_ppo" in r["Kernel_Name"]]
# the tightest rollout (graph replay) whose layers are the default kernel
best = None
for a, b in zip(idx[:-1], idx[1:]):
    names = [r["Kernel_Name"] for r in rows[a:b]]
    if b - a < 40 or not any("linear_split16_kernel<4" in n for n in names):
        continue
    span = int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])
    if best is None or span < best[0]:
        best = (span, a, b)
lo, hi = best[1] + 1, best[2] + 3
out = []
prev_end = int(rows[lo - 1]["End_Timestamp"])
tot_k = tot_g = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%-70s %8.2f us  gap %6.2f us" % (r["Kernel_Name"][:70], (e - s) / 1e3, (s - prev_end) / 1e3))
    tot_k += e - s; tot_g += max(0, s - prev_end)
    prev_end = max(prev_end, e)
out.append("kernels %.1f us, gaps %.1f us over %d kernels" % (tot_k / 1e3, tot_g / 1e3, hi - lo))
open(R + "/gpurun_out/trace_gaps.txt", "w").write("\n".join(out) + "\n")
PY
rm -rf $R/gpurun_out/trace_gaps
tail -80 $R/gpurun_out/trace_gaps.txt
