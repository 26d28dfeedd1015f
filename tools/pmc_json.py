"""profiles/rN_pmc.json from the rocprofv3 --pmc passes of tools/pmc_profile.sh: per bench profile slot (the names
bench.py's hipEvent timing uses) the measured HBM bytes per launch and the issue-slot accounting of the kernel.

  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   FETCH_SIZE counts 64 B per 128-B request for wide streaming
                                                                reads on gfx950 (MI355X_MICROARCH.md, HBM section); KiB units
  issue_share          = (VALU instructions * 2.6 + MFMA instructions * C) / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
                         what a SIMD charges with >= 2 waves resident (tools/ubench/mfma_valu_mix.hip): 2.6 cycles per
                         wave64 VALU instruction, C = 12.5 per v_mfma_f32_16x16x32_f16 issued beside vector work (16
                         alone; the 32x32x16 instructions of stage 1 are counted as two) or 32 per v_mfma_f32_16x16x4_f32 (no co-execution); GRBM_GUI_ACTIVE sums the 8 XCDs
  valu_busy_share      = 4 * SQ_ACTIVE_INST_VALU / SIMD cycles,  mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles:
                         how long the SIMDs' two execution pipes were occupied; their sum is ~1.07 for the stage-1 kernels
                         (the pipes overlap by a few per cent only: that sum is the roof these kernels sit under)
  wave_wait_share      = SQ_WAIT_ANY / SQ_WAVE_CYCLES       (parked at s_waitcnt / s_barrier)
  wave_issue_stall_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (an instruction ready, the pipe not)

usage: python tools/pmc_json.py <dir with fp16/ and fp32/ pass directories> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]
# the launch shape the passes were taken on (written by tools/pmc_run.sh; bench.py attaches the per-launch numbers to launches of
# exactly this shape and to no other)
shape = {"images_per_launch": 16, "hp": 1088, "wp": 1920}
for _f in sorted(glob.glob(root + "/*/shape.json")):
    shape = json.load(open(_f))


def slot_of(kernel: str):
    k = kernel.replace("(anonymous namespace)::", "").replace("void ", "").replace("balf::", "").split("(")[0]
    m = re.match(r"stage1_kernel16<(\d)", k)
    if m:
        return "stage1_pool" if m.group(1) == "2" else "stage1_%s_branch" % ("grid" if m.group(1) == "0" else "block")
    m = re.match(r"stage1_kernel32<(\d)", k)                            # (exact-fp32 stage 1 since round 5, stage1_f32.h)
    if m:
        return "stage1_%s_branch" % ("grid" if m.group(1) == "0" else "block")
    m = re.match(r"stage2_kernel16<(\d)", k)
    if m:
        return "stage2_pool" if m.group(1) == "2" else "stage2_%s_branch" % ("grid" if m.group(1) == "0" else "block")
    if k.startswith("stage3_tail_kernel16"):
        return "stage3_pool"
    m = re.match(r"stage_cs_kernel16<(\d+), \d+, (\d)>", k)
    if m:
        st = [32, 64, 128, 256].index(int(m.group(1))) + 1
        return "stage%d_pool" % st if m.group(2) == "2" else "stage%d_%s_branch" % (st, "grid" if m.group(2) == "0" else "block")
    m = re.match(r"stage_branch_kernel<(\d+), \d+, (\d)>", k)          # (the exact-fp32 path, detector.hip)
    if m:
        return "stage%d_%s_branch" % ([32, 64, 128, 256].index(int(m.group(1))) + 1, "grid" if m.group(2) == "0" else "block")
    m = re.match(r"pool_kernel<(\d+)>", k)
    if m:
        return "stage%d_pool" % ([32, 64, 128].index(int(m.group(1))) + 1)
    m = re.match(r"se_(?:reduce_)?kernel<(\d+)>", k)
    if m:
        return "stage%d_se" % ([32, 64, 128, 256].index(int(m.group(1))) + 1)
    if k.startswith("head_kernel"):
        return "stage4_head"
    if re.match(r"nms_tile(15_vec)?_kernel", k):           # the detection path runs nms_tile15_vec_kernel (window 15)
        return "nms_tile"
    if "topk_select_kernel" in k:
        return "topk_select"
    return None


res = {}
for prec in ("fp16", "fp32"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))       # slot -> counter -> sum over kernels of avg per dispatch
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{root}/{prec}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            v = per_kernel[r["Kernel_Name"]][r["Counter_Name"]]
            v[0] += float(r["Counter_Value"]); v[1] += 1
    kernels = collections.defaultdict(list)
    for kname, ctrs in per_kernel.items():
        s = slot_of(kname)
        if s is None:
            continue
        kernels[s].append(kname.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0])
        for c, (tot, n) in ctrs.items():
            acc[s][c] += tot / n
    mfma_cycles = 12.5 if prec == "fp16" else 32.0
    slots = {}
    for s, c in sorted(acc.items()):
        d = {"kernels": sorted(set(kernels[s])),
             "hbm_bytes_per_launch": (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0,
             "fetch_kib_raw": c.get("FETCH_SIZE", 0.0), "write_kib": c.get("WRITE_SIZE", 0.0)}
        if c.get("SQ_WAVE_CYCLES"):
            simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
            # (stages 1-2 of the split-f16 path issue v_mfma_f32_32x32x16_f16: twice the MACs and cycles of a 16x16x32)
            # (and the fp32 stage-1 kernels v_mfma_f32_32x32x2_f32: 64 cycles against 32 for the 16x16x4)
            big = s.startswith(("stage1_", "stage2_")) if prec == "fp16" else s in ("stage1_grid_branch", "stage1_block_branch")
            mf = c.get("SQ_INSTS_MFMA", 0.0) * (2.0 if big and not s.endswith("_se") else 1.0)
            d.update({"waves": c["SQ_WAVES"], "valu_insts": c["SQ_INSTS_VALU"], "mfma_insts": c.get("SQ_INSTS_MFMA", 0.0),
                      "issue_share": (c["SQ_INSTS_VALU"] * 2.6 + mf * mfma_cycles) / simd_cycles,
                      "wave_wait_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                      "wave_issue_stall_share": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                      # execution-pipe occupancy per SIMD cycle: the vector ALU (4 cycles per wave64 instruction, 16 per
                      # transcendental; the counter is in quad-cycles) and the matrix pipe (16 cycles per 16x16x32 f16 MFMA)
                      "valu_busy_share": 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles,
                      "mfma_busy_share": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles,
                      "gpu_cycles": c["GRBM_GUI_ACTIVE"] / 8.0})
            # LDS and L2 activity (round 4): instructions, array-busy and bank-conflict cycles per SIMD-cycle-equivalent of the
            # CU's one LDS (cycles / CU cycles), L2 requests and hit rate, L1 -> L2 read requests (64 B each)
            cu_cycles = 256.0 * c["GRBM_GUI_ACTIVE"] / 8.0
            for key, name in (("SQ_INSTS_LDS", "lds_insts"), ("SQ_INSTS_VMEM", "vmem_insts"), ("SQ_INSTS_SALU", "salu_insts")):
                if key in c:
                    d[name] = c[key]
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
                d["valu_mfma_coexec_share"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / simd_cycles
            if "SQ_LDS_IDX_ACTIVE" in c:
                d["lds_array_busy_share"] = c["SQ_LDS_IDX_ACTIVE"] / cu_cycles
                d["lds_idx_active_cycles"] = c["SQ_LDS_IDX_ACTIVE"]
            if "SQ_LDS_BANK_CONFLICT" in c:
                d["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / cu_cycles
            if "SQ_ACTIVE_INST_LDS" in c:
                d["lds_inst_active_share"] = 4.0 * c["SQ_ACTIVE_INST_LDS"] / simd_cycles
            if "TCC_REQ_sum" in c:
                d["l2_requests"] = c["TCC_REQ_sum"]
                if c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0) > 0:
                    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            if "TCP_TCC_READ_REQ_sum" in c:
                d["l1_to_l2_read_requests"] = c["TCP_TCC_READ_REQ_sum"]
        slots[s] = d
    res[prec] = slots
json.dump({"command": "tools/pmc_profile.sh: per precision three rocprofv3 --kernel-trace --pmc passes of "
                      "`python3 bench.py --steps 1 --warmup 1 --batch-per-gpu 16 --cpu-images 0 --other-steps 0 --other-configs 0` "
                      "(1088x1920, 16 images per launch): SQ counters + GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE",
           "shape": shape, "definitions": __doc__, "slots": res}, open(out, "w"), indent=1)
print(out)
