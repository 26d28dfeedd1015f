#!/bin/bash
# PMC passes over tools/bench_greedy.py (development aid): tools/pmc_greedy.sh <outdir under gpurun_out>
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/$1"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/tools/bench_greedy.py" > /dev/null 2>"$out/pass$i.err"
  i=$((i+1))
done
python3 "$root/tools/pmc_summary.py" "$out" > "$out/summary.txt"
find "$out" -name "*.csv" -delete; find "$out" -name "*.db" -delete
