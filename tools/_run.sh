tools/pmc_profile.sh gpurun_out/r2_pmc_v4 gpurun_out/r2_pmc_v4.json > gpurun_out/pmc_profile.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_v4_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-images 0 --other-configs 0 --no-single-rank-collective > $GRAFT_REPO_ROOT/gpurun_out/r2_v4_stats_bench.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | tail -1 > gpurun_out/r2_v4_bench.json
cut -c1-400 gpurun_out/r2_v4_bench.json
ls gpurun_out/r2_v4_stats/*/ | head
