#!/bin/bash
# usage: prof_stats.sh <name> <python script and args...>  -- rocprofv3 kernel trace + per-kernel stats of a tool, top kernels printed
set -e
R=$GRAFT_REPO_ROOT
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$NAME -- python3 "$@" > $R/gpurun_out/prof_$NAME.log 2>&1
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$NAME/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
import shutil; shutil.copy(f[0], "gpurun_out/r03_${NAME}_kernel_stats.csv")
for r in rows[:14]:
    print("%-100s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
