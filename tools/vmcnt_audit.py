"""Audit of hand-counted `s_waitcnt vmcnt(N)` against the BUILT code (used by tests/test_build_invariants.py and by hand:
python tools/vmcnt_audit.py [libbalf_hip.so] [kernel-name regex]).

The float-input stage-1 kernels issue their loop's loads from inline asm and wait for them with hand-placed counted waits
(stage1_f16.h).  Two things the source cannot enforce are checked in the disassembly of the kernel's main loop, walked twice
so that loop-carried loads meet the waits of the next iteration:
  (a) every vector-memory LOAD is covered: before the first later instruction that touches its destination registers there
      is an `s_waitcnt vmcnt(N)` with N <= the number of vector-memory operations issued between the load and that wait
      (they retire in issue order on gfx950, so the load has then landed);
  (b) nothing touches the destination between the load and that wait -- in particular no compiler-inserted v_mov that
      copies the register before the data has arrived (found in the round-2 build of the grid kernel: the copy sat a whole
      loop iteration after the load and in front of the wait, correct only by timing).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", so], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"b{i}.co")
        open(part, "wb").write(blob[a:b])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        out.append(co)
    return out


def disassemble(so, name_re):
    """{mangled kernel name: [(label or None, mnemonic, operand text)]} for the kernels whose name matches."""
    rx = re.compile(name_re)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--symbolize-operands", co], check=True, capture_output=True,
                                 text=True).stdout
            cur = None
            for ln in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:$", ln)
                if m:
                    if re.fullmatch(r"L\d+", m.group(1)):
                        if cur is not None:
                            res[cur].append((m.group(1), None, None))
                    else:
                        cur = m.group(1) if rx.search(m.group(1)) else None
                        if cur is not None:
                            res[cur] = []
                    continue
                if cur is None or not ln.startswith("\t"):
                    continue
                body = ln.split("//")[0].strip()
                if not body:
                    continue
                parts = body.split(None, 1)
                res[cur].append((None, parts[0], parts[1] if len(parts) > 1 else ""))
    return res


def vregs(text):
    """the VGPR numbers an operand text names: v12, v[10:13]"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


def is_vmem(mn):
    return mn.startswith(("global_", "buffer_", "scratch_", "flat_"))


def main_loop(ins):
    """Instruction list of the LAST backward-branch loop that contains a counted wait (vmcnt(N), N > 0), in execution
    order starting at the branch target."""
    labels = {lab: i for i, (lab, mn, _) in enumerate(ins) if lab}
    best = None
    for i, (lab, mn, ops) in enumerate(ins):
        if mn and mn.startswith(("s_branch", "s_cbranch")):
            t = labels.get(ops.strip())
            if t is not None and t < i:
                body = [x for x in ins[t:i + 1] if x[1]]
                if any(x[1] == "s_waitcnt" and re.search(r"vmcnt\(([1-9]\d*)\)", x[2]) for x in body):
                    if best is None or len(body) > len(best):
                        best = body
    return best


def audit(ins):
    """-> (list of problems, number of loads checked, list of (N, younger ops) per covering wait)"""
    loop = main_loop(ins)
    if loop is None:
        return ["no loop with a counted vmcnt wait found"], 0, []
    seq = loop + loop                       # two iterations: loop-carried loads meet the next iteration's waits
    problems, checked, margins = [], 0, []
    n = len(loop)
    for i, (_, mn, ops) in enumerate(seq[:n]):
        if not (is_vmem(mn) and "load" in mn):
            continue
        dest = vregs(ops.split(",")[0])
        younger, covered = 0, False
        for j in range(i + 1, min(i + 1 + n, len(seq))):
            _, m2, o2 = seq[j]
            if m2 == "s_waitcnt":
                w = re.search(r"vmcnt\((\d+)\)", o2)
                if w and int(w.group(1)) <= younger:
                    covered = True
                    margins.append((int(w.group(1)), younger))
                    break
            elif is_vmem(m2):
                younger += 1
                if "load" in m2 and vregs(o2.split(",")[0]) & dest:
                    problems.append(f"{mn} {ops}: destination re-loaded by `{m2} {o2}` before any covering wait")
                    break
            if m2 != "s_waitcnt" and vregs(o2) & dest:
                problems.append(f"{mn} {ops}: `{m2} {o2}` touches the destination {younger} vector-memory operations "
                                f"after the load and BEFORE any wait that covers it")
                break
        else:
            problems.append(f"{mn} {ops}: no covering wait within one loop iteration")
        checked += 1
        if not covered and not problems:
            problems.append(f"{mn} {ops}: not covered")
    return problems, checked, margins


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "balf_amd", "libbalf_hip.so")
    rx = sys.argv[2] if len(sys.argv) > 2 else r"stage1_kernel16ILi[01]ELb0"
    for name, ins in disassemble(so, rx).items():
        p, c, m = audit(ins)
        print(name, f"{c} loads checked, waits (N, younger ops): {sorted(set(m))}")
        for x in p:
            print("   PROBLEM:", x)
