#!/bin/bash
# PMC passes over a short bench run (development aid).  Usage: tools/pmc_run.sh <outdir> "<counters>" ...
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pass$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch-per-gpu 8 --cpu-images 0 > /dev/null 2>$out/pass$i.err
  i=$((i+1))
done
