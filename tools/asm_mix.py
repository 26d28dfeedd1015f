"""Instruction mix per kernel of a hipcc -S listing (development aid)."""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = Counter(ins)
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    mf = g('v_mfma')
    trans = sum(v for k, v in c.items() if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', k))
    print(f"{name[-70:]:70s} mfma {mf:5d} valu {g('v_') - mf:6d} trans {trans:4d} ds {g('ds_'):4d} gload {g('global_load'):4d} "
          f"gstore {g('global_store'):4d} scratch {g('scratch_'):4d} wait {c.get('s_waitcnt', 0):4d} barrier {c.get('s_barrier',0):2d} total {len(ins)}")
