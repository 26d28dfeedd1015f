import re, sys
txt = open(sys.argv[1]).read()
for b in re.split(r'\n(?=\S)', txt):
    lines = b.strip().split('\n'); name = lines[0]
    if 'stage_branch' not in name and 'head' not in name: continue
    d = {l.split()[0]: float(l.split()[1]) for l in lines[1:]}
    wc = d['SQ_WAVE_CYCLES']; w = d.get('SQ_WAVES', 0) or 1
    g = lambda k: d.get(k, 0.0)
    print(f"{name:34s} cyc/wave {wc*4/w:8.0f} mfma/w {g('SQ_INSTS_MFMA')/w:6.0f} valu/w {g('SQ_INSTS_VALU')/w:6.0f} vmem/w {g('SQ_INSTS_VMEM')/w:5.0f} "
          f"wait_any {g('SQ_WAIT_ANY')/wc:.2f} wait_inst {g('SQ_WAIT_INST_ANY')/wc:.2f} active {g('SQ_ACTIVE_INST_ANY')/wc:.2f} act_valu {g('SQ_ACTIVE_INST_VALU')/wc:.2f} "
          f"mfma_busy {g('SQ_VALU_MFMA_BUSY_CYCLES')/(wc*4):.2f} lds_wait {g('SQ_WAIT_INST_LDS')/wc:.2f}")
