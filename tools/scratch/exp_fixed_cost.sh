#!/bin/bash
# per-k-step slope / intercept of the layer kernel for every libmms_*.so build (timing experiments) and the default
L=$PWD/massive_marl_benchmark_amd/lib
: > gpurun_out/exp_fixed_cost.txt
for so in $L/libmms.so $L/libmms_*.so; do
  case $so in *libmms_cpu.so) continue;; esac
  echo "== $(basename $so)" >> gpurun_out/exp_fixed_cost.txt
  MMS_LIB=$so timeout -k 10 200 python tools/scratch/split16_fixed_cost.py 2>/dev/null | sed 's/K 32:.*K 2048: [0-9.]* us |//' >> gpurun_out/exp_fixed_cost.txt || exit 1
done
cat gpurun_out/exp_fixed_cost.txt
