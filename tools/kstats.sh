#!/bin/bash
# Static per-kernel figures of a gfx950 assembly file (development aid): tools/kstats.sh file.s [name filter]
# registers / spills from the kernel metadata, instruction-class counts from the body.
f=$1; pat=${2:-.}
python3 - "$f" "$pat" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
meta = {}
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    d = dict(re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_count|agpr_count|private_segment_fixed_size):\s+(\d+)", m.group(2)))
    meta[m.group(1)] = d
for name, d in meta.items():
    if not pat.search(name):
        continue
    b = re.search(r"^%s:[^\n]*\n(.*?)s_endpgm" % re.escape(name), txt, re.S | re.M)
    body = b.group(1) if b else ""
    ins = re.findall(r"^\s+([a-z_0-9]+)", body, re.M)
    c = lambda p: sum(1 for i in ins if re.match(p, i))
    print(f"{name[:70]:70s} vgpr {d.get('vgpr_count')} agpr {d.get('agpr_count')} spill {d.get('vgpr_spill_count')} scratch {d.get('private_segment_fixed_size')} | "
          f"valu {c(r'v_(?!mfma)')} mfma {c(r'v_mfma')} trans {c(r'v_(exp|rcp|rsq|log|sqrt)')} pk {c(r'v_pk_')} ds {c(r'ds_')} vmem {c(r'(global|buffer)_')} "
          f"salu {c(r's_(?!waitcnt|nop|barrier)')} waitcnt {c(r's_waitcnt')} nop {c(r's_nop')} barrier {c(r's_barrier')}")
PY
