tools/pmc_profile.sh gpurun_out/r2_pmc profiles/r2_pmc.json
ls gpurun_out/r2_pmc/fp16 gpurun_out/r2_pmc/fp32 | head; tail -2 gpurun_out/r2_pmc/fp16/pass0.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-images 0 --other-configs 0 --no-single-rank-collective > $GRAFT_REPO_ROOT/gpurun_out/r2_stats_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT; find gpurun_out/r2_stats -name "*kernel_stats.csv" | head -2
