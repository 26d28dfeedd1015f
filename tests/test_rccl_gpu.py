"""RCCL on the box we have (SURVEY.md 8e caveat): a single-rank ``nccl`` process group, created in a fresh child
process before any other GPU call, runs the path's one collective -- ``pipeline.allgather_keypoints`` with the
single-rank early return bypassed -- on device tensors and must hand them back unchanged.  2/4/8-GPU runs are the
driver's (SCALE_rNN.json); this test pins the plumbing: RCCL initialises, the [B, 2K+1] int32 slab packs/unpacks,
``all_gather_into_tensor`` executes on the stream."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from balf_amd import pipeline
g = torch.Generator().manual_seed(7)
b, k = 32, 2000
idx = torch.randint(0, 1080 * 1920, (b, k), generator=g, dtype=torch.int32).to(dev)
sc = torch.rand((b, k), generator=g).to(dev)
cnt = torch.randint(0, k + 1, (b,), generator=g, dtype=torch.int32).to(dev)
a, s, c = pipeline.allgather_keypoints(idx, sc, cnt, force=True)
torch.cuda.synchronize()
assert a.data_ptr() != idx.data_ptr(), "the collective was skipped"
assert torch.equal(a, idx) and torch.equal(s.view(torch.int32), sc.view(torch.int32)) and torch.equal(c, cnt)
# and the un-forced call keeps its single-rank shortcut
a2, _, _ = pipeline.allgather_keypoints(idx, sc, cnt)
assert a2 is idx
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK", torch.cuda.nccl.version())
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_env(**kw):
    """the RCCL environment of a rank comes from ONE place, bench.rank_environment (imports nothing GPU-related)"""
    sys.path.insert(0, ROOT)
    import bench
    return bench.rank_environment(dict(os.environ, **kw))


def test_allgather_keypoints_single_rank_rccl():
    env = _rank_env(MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") and env["MASTER_ADDR"]
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _device_count():
    """GPUs of this box WITHOUT touching HIP: torch.cuda.device_count() falls back to hipGetDeviceCount -- which
    initialises the runtime in the pytest process -- when amdsmi is not importable, and every later subprocess would then be the
    child of a GPU-initialised parent.  The KFD topology lists one node per agent; GPU nodes have a non-zero simd_count."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(base):
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(base, d, "properties")) if len(ln.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    visible = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    if visible is not None and visible.strip() != "":
        n = min(n, len([v for v in visible.split(",") if v.strip() != ""]))
    return n


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (the build's gpurun boxes have one; an 8-GPU node runs it)")
def test_allgather_two_ranks():
    """The first REAL N > 1 step: `python bench.py --gpus 2` -- the driver's command form -- starts two
    fresh ranks from a parent that never touches the GPU; each runs forward + NMS + top-K on its shard and the RCCL
    all-gather over xGMI; bench.py itself verifies on every rank that the gathered slabs are identical and that the rank's own
    shard sits at its rows, and refuses to print a line otherwise."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--other-configs", "0", "--other-steps", "0", "--cpu-images", "0"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["launched_by"] == "bench.py"
    assert res["gathered_slabs_identical"] is True and res["config"]["global_batch"] == 64
    assert res["rank_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert res["value"] > 0 and res["per_rank_images_per_s"]["min"] > 0


def test_two_ranks_on_one_gpu_rehearsal():
    """Everything the N = 2 step of bench.py does EXCEPT RCCL, on a one-GPU box (BALF_BENCH_REHEARSAL=1: both ranks on cuda:0,
    gloo through host memory): the launcher starts two ranks, each runs forward + NMS + top-K on ITS shard of the synthetic
    batch (images rank * b ...), the slabs are gathered, every rank verifies that all ranks hold the same gathered slabs with
    its own shard at its rows, the time is the maximum over the ranks, rank 0 prints one line -- marked as a rehearsal, never as
    a measurement.  The RCCL form of the same step waits for a box with two GPUs (test_allgather_two_ranks)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["BALF_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch-per-gpu", "3", "--height", "480", "--width", "640", "--topk", "1000",
                        "--other-configs", "0", "--other-steps", "0", "--cpu-images", "0"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["rehearsal"] is True and res["metric"].startswith("REHEARSAL")
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["launched_by"] == "bench.py"
    assert res["gathered_slabs_identical"] is True and res["config"]["global_batch"] == 6
    assert res["keypoints_per_image"] == 1000.0 and res["per_rank_images_per_s"]["min"] > 0


def test_four_ranks_on_one_gpu_rehearsal_at_the_real_shard():
    """VERDICT r5 item 6, within what a one-GPU box allows (at most 6 processes may hold the card, and this pytest process is one
    of them): FOUR ranks, each with the real per-GPU workload of BASELINE configs[3] -- 32 x 1080p, its own 14.8 GB workspace --
    global batch 128, gloo through host memory.  Every rank verifies inside bench.py that all ranks hold identical gathered slabs
    with its own shard at its rows; the line lists, per rank, the LOCAL_RANK it was given (the device it binds on a real node),
    the shard it computed and the workspace it sized.  A rehearsal, never a measurement; the 8-rank form of the launcher runs
    on the CPU in tests/test_bench_launcher.py."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["BALF_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--other-configs", "0", "--other-steps", "0", "--cpu-images", "0"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["rehearsal"] is True and res["n_gpus"] == 4 and res["rccl_ranks"] == 4
    assert res["gathered_slabs_identical"] is True and res["config"]["global_batch"] == 128
    assert res["keypoints_per_image"] == 2000.0
    devs = res["rank_devices"]
    assert [d["rank"] for d in devs] == [0, 1, 2, 3] and [d["local_rank_env"] for d in devs] == [0, 1, 2, 3]
    assert [d["images"] for d in devs] == [[0, 32], [32, 64], [64, 96], [96, 128]]
    assert all(14.0e9 < d["forward_workspace_bytes"] < 15.5e9 for d in devs), devs
