"""The HOST side of libbalf_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer:
weight packer, workspace planners, argument checks, state-tensor table, status plumbing -- everything tests/test_abi.py and
tests/test_host_api.py drive without a GPU -- run against a separately built library (balf_amd/csrc/build_asan.sh:
host code instrumented, device code as usual; GPU sanitizers are not available on this pool) in a child process that preloads
the sanitizer runtime.  A heap overflow in pack_frags*, a misaligned or out-of-range access in make_plan or a signed overflow in
an argument check aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, "balf_amd", "libbalf_hip_asan.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _runtime():
    r = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isfile(p) else None


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason="needs hipcc to build the instrumented library")
def test_host_entry_points_under_asan_ubsan():
    rt = _runtime()
    if rt is None:
        pytest.skip("the toolchain ships no shared AddressSanitizer runtime")
    src = os.path.join(ROOT, "balf_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".hip", ".h")))
    if not os.path.isfile(ASAN_LIB) or os.path.getmtime(ASAN_LIB) < newest:
        subprocess.check_call(["bash", os.path.join(src, "build_asan.sh")])
    out = subprocess.run(["ldd", ASAN_LIB], capture_output=True, text=True).stdout
    assert "libclang_rt.asan" in out, "the instrumented library does not link the sanitizer runtime"
    env = dict(os.environ, LD_PRELOAD=rt, BALF_HIP_LIB=ASAN_LIB,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_abi.py"), os.path.join(ROOT, "tests", "test_host_api.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout, tail
