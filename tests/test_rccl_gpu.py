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


def test_allgather_keypoints_single_rank_rccl():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
