#!/bin/bash
# The forward on random and on ZERO operands (weights and image), back to back for a few seconds each, with the shader clock and
# the package power sampled beside it (development aid): same library, same instruction streams -- what changes is
# the energy per instruction, and with it the clock the package can hold.  Usage: tools/power_zero.sh [seconds]
root="$(cd "$(dirname "$0")/.." && pwd)"
secs=${1:-6}
cd "$root"
for mode in rand zero; do
python3 - "$secs" "$mode" <<'PY' &
import os, sys, time, torch
sys.path.insert(0, ".")
os.environ["BALF_FP16_CHECK"] = "0"
from balf_amd import arch
from balf_amd.model import get_model
from balf_amd.utils import synth
mode = sys.argv[2]
sd = synth.synthetic_state_dict(1)
if mode == "zero":
    sd = {k: torch.zeros_like(v) if v.is_floating_point() else v for k, v in sd.items()}
    sd["detector_head.norm.running_var"] = torch.ones_like(sd["detector_head.norm.running_var"])
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(sd); m = m.eval().cuda()
x = torch.rand((16, 3, 1088, 1920), device="cuda") if mode == "rand" else torch.zeros((16, 3, 1088, 1920), device="cuda")
for _ in range(3): m(x, want_logits=False)
torch.cuda.synchronize()
t0 = time.time(); t_end = t0 + float(sys.argv[1]) + 3; n = 0
while time.time() < t_end:
    for _ in range(10): m(x, want_logits=False)
    torch.cuda.synchronize(); n += 10
dt = time.time() - t0
print(f"{mode}: {16 * n / dt:.1f} images/s ({m.effective_precision})")
PY
pid=$!
sleep 4
for i in $(seq 1 $((secs))); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/GPU\[0\]\s*: //g' | tr '\n' ' ' | sed -e "s/^/$mode: /"
  echo
  sleep 1
done
wait $pid
done
