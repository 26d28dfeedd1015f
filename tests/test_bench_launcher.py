"""bench.py's N > 1 launch path on a box without GPUs: `python bench.py --gpus 2` starts two fresh
ranks itself, the parent never imports torch or the product library, rank 0 prints ONE JSON line with n_gpus 2, and
every way of ending up with fewer devices or ranks than --gpus fails loudly instead of reporting n_gpus = 1.
The step is the stub of bench.stub_main (gloo, fake keypoint slabs): this tests plumbing, it measures nothing."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
STUB = ["--stub-step", "--steps", "3", "--warmup", "1", "--batch-per-gpu", "4", "--topk", "16"]


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID")}
    env.update(kw)
    return env


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _one_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_plain_gpus2_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + STUB, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _one_line(r.stdout)
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["stub"] is True
    assert res["launched_by"] == "bench.py" and res["config"]["global_batch"] == 8
    assert res["per_rank_images_per_s"]["min"] <= res["per_rank_images_per_s"]["max"]


def test_plain_gpus8_starts_eight_ranks():
    """the node size the driver's scaling run ends with (N = 8): eight ranks, one line, global batch 8 x per-rank batch"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"] + STUB, env=_env(OMP_NUM_THREADS="1"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _one_line(r.stdout)
    assert res["n_gpus"] == 8 and res["rccl_ranks"] == 8 and res["config"]["global_batch"] == 32
    assert res["config"]["parallelism"] == "dp8" and res["scaling"] == "weak"
    # every rank got its own LOCAL_RANK (the device it binds on a real node) and its own contiguous shard of the batch
    devs = res["rank_devices"]
    assert [d["rank"] for d in devs] == list(range(8)) and [d["would_bind"] for d in devs] == [f"cuda:{r}" for r in range(8)]
    assert [d["images"] for d in devs] == [[4 * r, 4 * r + 4] for r in range(8)]


def test_torchrun_form_gives_the_same_line():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2"] + STUB,
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _one_line(r.stdout)
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["launched_by"] == "external launcher"


def test_world_size_mismatch_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + STUB, env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "--gpus 2 but WORLD_SIZE=1" in r.stderr


def test_too_few_devices_fails_loudly():
    """the real step on this GPU-less box: every rank sees 0 devices, exits non-zero, and the launcher prints no line"""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "refusing to report" in r.stderr


def test_launcher_parent_does_not_load_torch_or_the_library():
    """the parent of `--gpus N` must not touch the GPU: it may not even import torch or load libbalf_hip.so"""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2'] + %r\n"
            "import subprocess\n"
            "class P:\n"
            "    def __init__(s, *a, **k): s.stdout = open('/dev/null', 'rb')\n"
            "    def poll(s): return 7\n"
            "    def terminate(s): pass\n"
            "subprocess.Popen = P\n"
            "try:\n"
            "    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    assert e.code == 7, e.code\n"
            "assert 'torch' not in sys.modules and 'balf_amd' not in sys.modules and 'numpy' not in sys.modules\n"
            "print('clean')\n") % (STUB, BENCH)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "clean", r.stderr[-2000:]


def test_ranks_get_the_rccl_environment_from_one_place():
    """the multi-rank RCCL group needs dmabuf IPC on this pool (HSA_ENABLE_IPC_MODE_LEGACY=0); the launcher,
    a rank under another launcher and the GPU test's child all take it from bench.rank_environment."""
    env = _env()
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + STUB, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_line(r.stdout)["rank_env"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2"] + STUB,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_line(r.stdout)["rank_env"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    # an explicit setting of the caller's wins
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + STUB, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and _one_line(r.stdout)["rank_env"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}


def test_a_rank_that_ignores_sigterm_is_killed():
    """after a rank fails the launcher terminates the others, and a rank that does not die on SIGTERM (stuck in
    a collective or a driver call) is killed after a bounded grace period instead of being polled forever."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3"] + STUB,
                       env=_env(BALF_BENCH_TEST_FAULT="hang-after-failure", BALF_BENCH_GRACE_S="2"), capture_output=True,
                       text=True, timeout=120)
    took = time.monotonic() - t0
    assert r.returncode == 3 and r.stdout.strip() == "", (r.returncode, r.stdout, r.stderr[-1500:])
    assert "ignored SIGTERM" in r.stderr and took < 60, (took, r.stderr[-1500:])


def test_the_whole_run_has_a_deadline():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + STUB,
                       env=_env(BALF_BENCH_TEST_FAULT="hang-all", BALF_BENCH_DEADLINE_S="3", BALF_BENCH_GRACE_S="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == "" and "did not finish within" in r.stderr, r.stderr[-1500:]


def test_global_batch_that_does_not_divide():
    """7 images on 3 ranks.  The stub step runs the padded collective of pipeline.allgather_keypoints(total=)
    (shards of 3 + 2 + 2) and every rank checks its own rows in the gathered slab; the MEASURED step refuses such a batch."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--global-batch", "7"] + STUB, env=_env(OMP_NUM_THREADS="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = _one_line(r.stdout)
    assert res["n_gpus"] == 3 and res["config"]["global_batch"] == 7
    # the measured step: one rank, WORLD_SIZE 2 in the environment, a batch of 5 -> refused before any GPU work
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--global-batch", "5"],
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not divide" in r.stderr, r.stderr[-2000:]
