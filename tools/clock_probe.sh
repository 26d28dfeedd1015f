#!/bin/bash
# Sample GPU clock / power while the forward runs back to back (development aid): tools/clock_probe.sh [seconds] [fp16|fp32]
root="$(cd "$(dirname "$0")/.." && pwd)"
secs=${1:-8}
prec=${2:-fp16}
python3 - "$secs" "$prec" <<'PY' &
import os, sys, time, torch
sys.path.insert(0, ".")
os.environ["BALF_FP16_CHECK"] = "0"      # (a timing-ablation library fails the split-f16 range check and would be run on the fp32 kernels)
from balf_amd import arch
from balf_amd.model import get_model
from balf_amd.utils import synth
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m.precision = sys.argv[2]; m = m.eval().cuda()
x = torch.rand((8, 3, 1088, 1920), device="cuda")
for _ in range(3): m(x, want_logits=False)
torch.cuda.synchronize()
t0 = time.time()
t_end = t0 + float(sys.argv[1]) + 3
n = 0
while time.time() < t_end:
    for _ in range(20): m(x, want_logits=False)
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
print(f"forwards: {n} in {dt:.2f} s = {8 * n / dt:.1f} images/s ({m.effective_precision})")
PY
pid=$!
sleep 3
for i in $(seq 1 $((secs * 2))); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' | sed -e 's/GPU\[0\]\s*: //g'
  echo
  sleep 0.5
done
wait $pid
