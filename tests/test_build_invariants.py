"""Invariants of the BUILT library that the source cannot express (no GPU needed): read the gfx950 code objects out of
libbalf_hip.so and check kernel metadata.

The float-input stage-1 kernels issue their loads from inline asm with hand-counted `s_waitcnt vmcnt(N)` (stage1_f16.h:
the compiler would otherwise drain the store queue at every loop back edge).  Scratch traffic is vector-memory traffic:
a register spill inside that loop shifts every count and the kernel computes on data that has not arrived (seen when
the block kernel was forced to 168 registers).  So those kernels must not spill."""
import os
import re
import shutil
import subprocess

import pytest

from balf_amd import _lib

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernel_metadata(tmp_path):
    for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, t)):
            pytest.skip(f"{t} not available")
    so = _lib.LIB_PATH
    fat = tmp_path / "fat.bin"
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", so], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    meta = {}
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part, co = tmp_path / f"bundle{i}.bin", tmp_path / f"bundle{i}.co"
        part.write_bytes(blob[a:b])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        # (the note's keys are sorted: .group_segment_fixed_size precedes .name, the register counts follow it)
        for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+)\n((?:(?!\.group_segment_fixed_size).)*?)\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, re.S):
            d = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", m.group(4))}
            d["group_segment_fixed_size"] = int(m.group(1))
            meta[m.group(3)] = d
    return meta


def test_asm_load_kernels_do_not_spill(tmp_path):
    if shutil.which("c++filt") is None:
        pytest.skip("c++filt not available")
    meta = _kernel_metadata(tmp_path)
    names = [n for n in meta if "stage1_kernel16" in n or "stage2_kernel16" in n]
    assert names, "persistent stage kernels not found in the library"
    checked = 0
    for n in names:
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        m = re.search(r"stage1_kernel16<(\d), (true|false)>", dem)
        m2 = re.search(r"stage2_kernel16<(\d)>", dem)
        assert m or m2, dem
        if (m and m.group(2) == "false" and m.group(1) in ("0", "1")) or (m2 and m2.group(1) in ("0", "1")):   # counted waits
            assert meta[n]["vgpr_spill_count"] == 0 and meta[n]["private_segment_fixed_size"] == 0, (dem, meta[n])
            checked += 1
    assert checked == 4


def test_hand_counted_vmcnt_waits_cover_their_loads():
    """ADVICE r2 / DESIGN 4.3e: in the built code of the two float-input stage-1 kernels every load of the main loop is
    covered by a counted wait before anything touches its destination, and nothing (no compiler-inserted copy) reads the
    destination between the load and that wait.  The round-2 build FAILED this check: hipcc copied the prefetched pixels
    out of the load's destination register in front of the wait (tools/vmcnt_audit.py)."""
    import importlib.util
    for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"):
        if not os.path.exists(os.path.join(LLVM, t)):
            pytest.skip(f"{t} not available")
    spec = importlib.util.spec_from_file_location(
        "vmcnt_audit", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "vmcnt_audit.py"))
    va = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(va)
    kernels = va.disassemble(_lib.LIB_PATH, r"stage1_kernel16ILi[01]ELb0")
    assert len(kernels) == 2, list(kernels)
    for name, ins in kernels.items():
        problems, checked, margins = va.audit(ins)
        assert not problems, (name, problems)
        grid = "ILi0E" in name
        assert checked == (4 if grid else 10), (name, checked)         # 4 pixel loads | 8 u' fragments + 2 pixel loads
        assert all(young >= n for n, young in margins), margins
        assert {n for n, _ in margins} == ({8} if grid else {2, 8}), margins
    # the stage-2 kernels (stage2_f16.h): grid = 4 input-fragment loads behind 8 u' stores; block = 4 chunks of 8 weight
    # loads + 8 u' fragments + 4 input fragments, waits at 12 / 12 / 8 / 0
    kernels = va.disassemble(_lib.LIB_PATH, r"stage2_kernel16ILi[01]E")
    assert len(kernels) == 2, list(kernels)
    for name, ins in kernels.items():
        problems, checked, margins = va.audit(ins)
        assert not problems, (name, problems)
        grid = "ILi0E" in name
        assert checked == (4 if grid else 44), (name, checked)
        assert all(young >= n for n, young in margins), margins
        assert {n for n, _ in margins} == ({8} if grid else {0, 8, 12}), margins
    # the stage-2 tail prefetches x1 (8 loads) and the input fragments (4) a group ahead; the compiler's own 8 loads of the
    # squeeze-excite scale follow them in the queue, and its counted waits for those cover the prefetch as well
    (name, ins), = va.disassemble(_lib.LIB_PATH, r"stage2_kernel16ILi2E").items()
    problems, checked, margins = va.audit(ins)
    assert not problems and checked == 20 and all(young >= n for n, young in margins), (name, problems, checked, margins)


def test_table_gelu_kernels_have_no_static_lds(tmp_path):
    """The table-GELU kernels (stage1_f16.h: gelu_lut_n, stage_cs_f16.h: gelu_log_n) use the masked index bits as the LDS
    ADDRESS of the table entry: the table is the first region of the dynamic LDS block, which starts at LDS address 0 only
    while the kernel has no static __shared__ allocation (those come first).  Also: none of them may spill -- a spilled
    address or pair register between the asm blocks would be a scratch access the hand-placed waits do not cover."""
    meta = _kernel_metadata(tmp_path)
    names = [n for n in meta if "stage1_kernel16" in n or "stage2_kernel16" in n or "stage_cs_kernel16" in n or "stage3_tail_kernel16" in n
             or "stage1_kernel32" in n]                    # (the exact-fp32 stage-1 kernels read the same table the same way, stage1_f32.h)
    assert len(names) >= 6 + 3 + 4 + 1 + 2, names          # stage 1: 3 modes x 2 inputs; stage 2: 3 modes; stages 3-4: 2 + 2; stage-3 tail; fp32 stage 1
    for n in names:
        assert meta[n]["group_segment_fixed_size"] == 0, (n, meta[n])
        assert meta[n]["vgpr_spill_count"] == 0, (n, meta[n])


def test_no_inline_asm_valu_writes_fresh_register():
    """Source lint.  hipcc pads the MFMA hazards (a VALU write within a few wait states of an MFMA that still reads the
    register as SrcC corrupts the accumulator input) for its own instructions only.  An inline-asm VALU instruction must
    therefore write the register of one of its own live inputs ("+v"), never a fresh output ("=v" / "=&v"), which the
    register allocator may place on the SrcC of an MFMA the scheduler hoisted in front of it (split16.h, DESIGN 4.3b/4.3d:
    found twice as rare non-deterministic errors)."""
    src = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")
    bad = []
    for name in sorted(os.listdir(src)):
        if not name.endswith((".h", ".hip")):
            continue
        text = open(os.path.join(src, name)).read()
        for m in re.finditer(r'\basm\s*(?:volatile)?\s*\(\s*((?:"(?:[^"\\]|\\.)*"\s*)+):([^:;]*)', text):
            template, outputs = m.group(1), m.group(2)
            if re.search(r'"\s*v_', template) is None and re.search(r'\\t\s*v_', template) is None:
                continue                                    # no vector-ALU instruction in this statement
            if re.search(r'"=&?v"', outputs):
                bad.append((name, text[:m.start()].count("\n") + 1, template.strip()[:60]))
    assert not bad, bad


def _audit_module():
    import importlib.util
    for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"):
        if not os.path.exists(os.path.join(LLVM, t)):
            pytest.skip(f"{t} not available")
    spec = importlib.util.spec_from_file_location(
        "vmcnt_audit", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "vmcnt_audit.py"))
    va = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(va)
    return va


def _regs(operand):
    """'v7' -> {7}, 'v[8:9]' -> {8, 9}, anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def test_shipped_library_is_a_release_build():
    """the timing-ablation / instrumentation switches (csrc/diag.h: wrong results) must be provably off in the
    library the tests and the bench load.  balf_build_flags() is compiled from the same macros the kernels see."""
    flags = _lib.lib().balf_build_flags().decode()
    assert flags.startswith("release "), flags
    items = dict(kv.split("=") for kv in flags.split()[1:])
    assert len(items) >= 10 and all(k.startswith("BALF_") for k in items), flags
    assert all(v == "0" for v in items.values()), flags
    for must in ("BALF_ABLATE_BARRIER", "BALF_ABLATE_LUTCOPY", "BALF_ABLATE_WSTREAM", "BALF_ABLATE_GELU", "BALF_DROP_WLO", "BALF_STAMPS"):
        assert must in items, (must, flags)
    # and the source keeps them in ONE place: no other header or translation unit defines such a switch
    src = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".h", ".hip")) and name != "diag.h":
            text = open(os.path.join(src, name)).read()
            assert not re.search(r"#\s*define\s+(BALF_ABLATE_\w+|BALF_DROP_WLO|BALF_STAMPS|BALF_HN_STAMPS|BALF_S1_STRICT)\b", text), name


def test_se_channel_sums_add_both_results_of_their_row_swap():
    """DESIGN 4.3e mis-fold #1, as a code-object invariant.  The channel sums of the RCAB's hidden layer (stage1_f16.h, the
    block kernel) add the TWO results of a v_permlane16_swap; hipcc once folded the second into the first -- the built code
    read `v_pk_add_f32 v[8:9], v[0:1], v[0:1]`, the sums came out as 2 * sw[0] and the squeeze-excite scales a few per cent
    off (inside the goldens' tolerance on the first run).  In the built block kernels every swap pair (vA, vB) must be consumed
    by an add that reads BOTH registers, and no add may read one register as both of its sources."""
    va = _audit_module()
    kernels = va.disassemble(_lib.LIB_PATH, r"stage1_kernel16ILi1ELb[01]")
    assert len(kernels) == 2, list(kernels)
    # the stage-2 block kernel (stage2_f16.h) has the same reduction over twice the registers; its first build summed
    # register 0 sixteen times (__builtin_bit_cast applied to a vector element reads element 0: DESIGN 4.3f) -- the built code
    # then held TWO swaps instead of sixteen
    k2 = va.disassemble(_lib.LIB_PATH, r"stage2_kernel16ILi1E")
    assert len(k2) == 1, list(k2)
    kernels.update(k2)
    for name, ins in kernels.items():
        swaps = [(i, ops) for i, (_, mn, ops) in enumerate(ins) if mn and mn.startswith("v_permlane16_swap")]
        assert len(swaps) == (16 if "stage2" in name else 8), (name, len(swaps))   # one per register pair (r, r + 8) of a tile's sums
        for i, ops in swaps:
            a, b = (_regs(o.strip()) for o in ops.split(","))
            assert a and b and not (a & b), (name, ops)
            a, b = set(a), set(b)
            for _, mn, o2 in ins[i + 1:i + 200]:
                if mn and mn.startswith("v_mov_b32"):
                    # hipcc may copy a result next to its partner so that two sums pack into one v_pk_add_f32: the copy carries
                    # the value (its destination joins the set; anything else written there leaves it)
                    dst, src = (_regs(t.strip()) for t in o2.split(",")[:2])
                    for grp in (a, b):
                        if src and src & grp:
                            grp |= dst
                        elif dst:
                            grp -= dst
                    continue
                if not mn or not re.match(r"v_(pk_)?add_f32", mn):
                    continue
                srcs = [_regs(s.strip()) for s in o2.split(",")[1:3]]
                if not any(a & s for s in srcs):
                    continue
                assert any(a & s for s in srcs) and any(b & s for s in srcs), (name, ops, mn, o2)
                assert srcs[0] != srcs[1], (name, mn, o2)          # x + x: the mis-fold's signature
                break
            else:
                raise AssertionError(f"{name}: no add consumes the swap {ops}")
        for _, mn, o2 in ins:
            if mn and re.match(r"v_(pk_)?add_f32", mn):
                s = [t.strip() for t in o2.split(",")]
                assert len(s) < 3 or s[1] != s[2] or not _regs(s[1]), (name, mn, o2)


def test_tail_kernel_pools_sixteen_distinct_values():
    """DESIGN 4.3e mis-fold #2.  The stage-1 tail kernel brings the vertical partner's maximum into every lane with two DPP
    moves per accumulator register (row_shl:4, then row_shr:4 into lane banks 1 and 3); a first form let hipcc treat the
    sixteen results as one value and delete fifteen of the sixteen maxima.  The built kernels must hold sixteen moves of
    each kind, each reading a freshly computed value."""
    va = _audit_module()
    kernels = va.disassemble(_lib.LIB_PATH, r"stage1_kernel16ILi2ELb[01]")
    assert len(kernels) == 2, list(kernels)
    for name, ins in kernels.items():
        for pat in (r"row_shl:4 row_mask:0xf bank_mask:0xf", r"row_shr:4 row_mask:0xf bank_mask:0xa"):
            idx = [i for i, (_, mn, ops) in enumerate(ins) if mn == "v_mov_b32_dpp" and pat in ops]
            assert len(idx) == 16, (name, pat, len(idx))
            # sixteen different VALUES: identify what a move reads by the instruction that last wrote its source register
            # (the compiler re-uses a few registers for the sixteen maxima)
            defs = set()
            for i in idx:
                src = _regs(ins[i][2].split(",")[1].split()[0])
                d = next((j for j in range(i - 1, -1, -1)
                          if ins[j][1] and ins[j][2] and _regs(ins[j][2].split(",")[0].strip()) & src), None)
                assert d is not None, (name, pat, i)
                defs.add(d)
            assert len(defs) == 16, (name, pat, sorted(defs))


def test_the_vmcnt_audit_itself_catches_what_it_is_for():
    """The audit of the hand-counted waits (tools/vmcnt_audit.py) on synthetic loops: it accepts a correct schedule and names (a) a
    wait whose count is larger than the number of younger vector-memory operations, (b) a compiler-style copy of the load's
    destination in front of the wait (the round-2 race), (c) a load that no wait covers within an iteration."""
    va = _audit_module()

    def loop(body):
        return [("L1", None, None)] + [(None, mn, ops) for mn, ops in body] + [(None, "s_cbranch_scc1", "L1")]
    store = ("global_store_dwordx4", "v[100:101], v[40:43], off")
    good = loop([("s_waitcnt", "vmcnt(2)"), ("v_add_f32_e32", "v9, v4, v5"),
                 ("global_load_dwordx4", "v[4:7], v20, s[2:3]"), store, store])
    problems, checked, margins = va.audit(good)
    assert not problems and checked == 1 and margins == [(2, 2)], (problems, checked, margins)
    # (a) the wait allows three operations in flight but only two are younger than the load: the load may still be one of them
    bad_count = loop([("s_waitcnt", "vmcnt(3)"), ("v_add_f32_e32", "v9, v4, v5"),
                      ("global_load_dwordx4", "v[4:7], v20, s[2:3]"), store, store])
    assert va.audit(bad_count)[0], "a wait that does not cover its load went unnoticed"
    # (b) the destination is copied in front of the wait
    bad_copy = loop([("v_mov_b32_e32", "v30, v4"), ("s_waitcnt", "vmcnt(2)"), ("v_add_f32_e32", "v9, v30, v5"),
                     ("global_load_dwordx4", "v[4:7], v20, s[2:3]"), store, store])
    p = va.audit(bad_copy)[0]
    assert p and "touches the destination" in p[0], p
    # (c) no counted wait at all behind the load
    no_wait = loop([("s_waitcnt", "vmcnt(2)"), ("global_load_dwordx4", "v[4:7], v20, s[2:3]"), store, store,
                    ("global_load_dwordx4", "v[12:15], v21, s[2:3]"), ("s_waitcnt", "lgkmcnt(0)")])
    assert va.audit(no_wait)[0]


def test_fp32_stage1_pair_sums_add_both_results_of_their_lane_swap():
    """The same mis-fold in the exact-fp32 stage-1 kernels (stage1_f32.h: f1_pair_sum): the LayerNorm statistics add the two
    results of a v_permlane32_swap; without the opaque pass-through hipcc emitted `v_add_f32 v17, v0, v0` (seen in the
    assembly, round 5: every LayerNorm of the kernel was wrong).  Every swap (vA, vB) of the built kernels must be consumed by
    an add that reads both registers."""
    va = _audit_module()
    kernels = va.disassemble(_lib.LIB_PATH, r"stage1_kernel32ILi[01]E")
    assert len(kernels) == 2, list(kernels)
    for name, ins in kernels.items():
        swaps = [(i, ops) for i, (_, mn, ops) in enumerate(ins) if mn and mn.startswith("v_permlane32_swap")]
        assert len(swaps) >= 12, (name, len(swaps))              # two per LayerNorm and tile
        for i, ops in swaps:
            a, b = (_regs(o.strip()) for o in ops.split(","))
            assert a and b and not (a & b), (name, ops)
            for _, mn, o2 in ins[i + 1:i + 60]:
                if not mn or not re.match(r"v_add_f32", mn):
                    continue
                srcs = [_regs(s.strip()) for s in o2.split(",")[1:3]]
                if not any((a | b) & s for s in srcs):
                    continue
                assert any(a & s for s in srcs) and any(b & s for s in srcs), (name, ops, mn, o2)
                break
            else:
                raise AssertionError(f"{name}: no add consumes the swap {ops}")


def test_fp32_kernels_do_not_spill(tmp_path):
    """The exact-fp32 kernels sit at their register targets (252 of 256 at C = 128, 442 of 512 at C = 256 with four weight steps in
    flight): the GEMM's ring depth (detector.hip: BALF_F32_D*) was chosen per stage so that none of them spills -- D = 3 at C = 128
    did (5 registers, 4 % slower, profiles/r5_f32.txt)."""
    meta = _kernel_metadata(tmp_path)
    names = [n for n in meta if "stage_branch_kernel" in n or ("head_kernel" in n and "16" not in n) or "stage1_kernel32" in n]
    assert len(names) >= 6 + 1 + 2, names                  # generic kernel: stages 2-4 x 2 modes; head; stage 1 x 2 modes
    for n in names:
        assert meta[n]["vgpr_spill_count"] == 0 and meta[n]["private_segment_fixed_size"] == 0, (n, meta[n])
