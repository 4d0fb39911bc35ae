cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base noout noin; do
  MMS_LIB=$R/massive_marl_benchmark_amd/lib/libmms_$v.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/var_$v -- python3 $R/tools/profile_step.py --steps 128 --rollout-outputs > /dev/null 2>&1
  echo "variant $v"; python3 $R/tools/pmc_summary.py $R/gpurun_out/var_$v | grep -A4 ant_step
done
