cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 1 2 3; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/sub$n -- python3 $R/tools/profile_step.py --steps 64 --substeps $n --rollout-outputs > /dev/null 2>&1
  echo "substeps $n"; python3 $R/tools/pmc_summary.py $R/gpurun_out/sub$n | grep -A4 ant_step
done
