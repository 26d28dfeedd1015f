#!/bin/bash
# Sample GPU clock / power while the forward runs back to back (development aid): tools/clock_probe.sh [seconds]
root="$(cd "$(dirname "$0")/.." && pwd)"
secs=${1:-8}
python3 - "$secs" <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from balf_amd import arch
from balf_amd.model import get_model
from balf_amd.utils import synth
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m = m.eval().cuda()
x = torch.rand((8, 3, 1088, 1920), device="cuda")
t_end = time.time() + float(sys.argv[1]) + 3
n = 0
while time.time() < t_end:
    for _ in range(20): m(x, want_logits=False)
    torch.cuda.synchronize(); n += 20
print("forwards:", n)
PY
pid=$!
sleep 3
for i in $(seq 1 $((secs * 2))); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' | sed -e 's/GPU\[0\]\s*: //g'
  echo
  sleep 0.5
done
wait $pid
