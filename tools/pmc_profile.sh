#!/bin/bash
# The committed PMC profile: tools/pmc_profile.sh <outdir under gpurun_out> <profiles/rN_pmc.json>
# Per precision: SQ instruction/wait counters + GRBM_GUI_ACTIVE, pipe-busy counters, FETCH_SIZE, WRITE_SIZE (separate passes, --kernel-trace only).
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$1; json=$2
for prec in fp16 fp32; do
  PREC=$prec "$root/tools/pmc_run.sh" "$out/$prec" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
    "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
    "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
    "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"
done
python3 "$root/tools/pmc_json.py" "$root/$out" "$root/$json"
