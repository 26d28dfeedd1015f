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
    names = [n for n in meta if "stage1_kernel16" in n]
    assert names, "stage-1 kernels not found in the library"
    checked = 0
    for n in names:
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        m = re.search(r"stage1_kernel16<(\d), (true|false)>", dem)
        assert m, dem
        if m.group(2) == "false" and m.group(1) in ("0", "1"):       # float input, counted waits
            assert meta[n]["vgpr_spill_count"] == 0 and meta[n]["private_segment_fixed_size"] == 0, (dem, meta[n])
            checked += 1
    assert checked == 2


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


def test_table_gelu_kernels_have_no_static_lds(tmp_path):
    """The table-GELU kernels (stage1_f16.h: gelu_lut_n, stage_cs_f16.h: gelu_log_n) use the masked index bits as the LDS
    ADDRESS of the table entry: the table is the first region of the dynamic LDS block, which starts at LDS address 0 only
    while the kernel has no static __shared__ allocation (those come first).  Also: none of them may spill -- a spilled
    address or pair register between the asm blocks would be a scratch access the hand-placed waits do not cover."""
    meta = _kernel_metadata(tmp_path)
    names = [n for n in meta if "stage1_kernel16" in n or "stage_cs_kernel16" in n]
    assert len(names) >= 6 + 8, names
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
