#!/bin/bash
# A/B over builds of the layer kernel: every libmms_<name>.so under massive_marl_benchmark_amd/lib + the default, two rounds
L=$PWD/massive_marl_benchmark_amd/lib
: > gpurun_out/roll_ab.jsonl
for round in 1 2; do
  for so in $L/libmms_*.so $L/libmms.so; do
    case $so in *libmms_cpu.so) continue;; esac
    MMS_LIB=$so timeout -k 10 200 python tools/scratch/split16_roll_ab.py >> gpurun_out/roll_ab.jsonl 2>> gpurun_out/roll_ab.err || exit 1
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/roll_ab.jsonl'):
    d=json.loads(l)
    print(d['lib'].split('/')[-1].ljust(24), d['ppo_hidden_sha'][:6], d['ppo_value_small_sha'][:6], d['marl_sha'][:6], '%.1f %.1f %.1f'%(d['ppo_both_us'],d['ppo_value_us'],d['marl_pass_us']))
PY
