"""The detection pipeline around the model: pad -> forward -> crop/border/NMS/top-K, batched and
data-parallel.

``extract_detections`` mirrors the reference's canonical caller
(/root/reference/balf/utils/train_utils.py:416-454): same arguments, same return values
(``pts_output`` rows ``[x, y, 1.0, score]`` float64 sorted by score, and the cropped score-map
batch).  ``detect_batch`` is the batched form the benchmark drives: everything stays on the GPU and
only ``[B, K]`` keypoint slabs come out.  ``allgather_keypoints`` is the one collective of the
multi-GPU path: images are independent, so ranks shard the batch and exchange
nothing but their fixed-size keypoint slabs (RCCL all-gather over xGMI; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import arch, ops
from .utils import test_utils as T


def detect_batch(model, x_pad: torch.Tensor, h: int, w: int, border: int = 15, nms_size: int = 15,
                 num_points: int = 1000, precomputed_offsets: Optional[Tuple[int, int]] = None, want_logits: bool = False):
    """x_pad [B,3,Hp,Wp] (already padded as ``mod_padding_symmetric`` does) -> (idx [B,K] int32 flat
    ``y*w+x`` in un-padded coordinates, score [B,K] fp32, count [B] int32, prob [B,Hp,Wp]).  ``want_logits``: the head kernel
    also writes the [B,65,Hp/8,Wp/8] logits the reference's forward always returns (nothing here reads them: +0.1 % of a step)."""
    hp, wp = x_pad.shape[-2:]
    if precomputed_offsets is None:
        ehp, ewp, top, left = arch.padded_hw(h, w)
        if (ehp, ewp) != (hp, wp):
            raise ValueError(f"a {h}x{w} image pads to {ehp}x{ewp}, got {hp}x{wp}")
    else:
        top, left = precomputed_offsets
    prob = model(x_pad, want_logits=want_logits)["prob"]
    idx, score, count = ops.nms_topk(prob, top, left, h, w, border, nms_size, num_points)
    return idx, score, count, prob


def detect_batch_u8(model, images_u8: torch.Tensor, border: int = 15, nms_size: int = 15, num_points: int = 1000):
    """Raw uint8 images on the GPU (gray ``[B,H,W]`` or RGB ``[B,H,W,3]``) -> keypoints, nothing on the host:
    normalisation and padding are fused into the first kernels (``MLP_MA_DECODER.forward_u8``), crop/border/NMS/
    top-K into the last.  Same return values as :func:`detect_batch`.

    Split-f16 guard: the NMS is enqueued right behind the forward, so under the default ``BALF_FP16_GUARD=lazy`` a batch whose
    status block turns out flagged has gone through the NMS with the split path's score map; ``model.fp16_guard_check()``
    returns True then (and repairs ``prob``): repeat the call, or run with ``BALF_FP16_GUARD=sync``."""
    h, w = images_u8.shape[1], images_u8.shape[2]
    _, _, top, left = arch.padded_hw(h, w)
    prob = model.forward_u8(images_u8, want_logits=False)["prob"]
    idx, score, count = ops.nms_topk(prob, top, left, h, w, border, nms_size, num_points)
    return idx, score, count, prob


class GraphedDetector:
    """:func:`detect_batch_u8` for ONE input shape, captured once into a hipGraph (``torch.cuda.graph``) and replayed per call.

    The library is stream-ordered end to end -- no entry point synchronises or reads anything back (round 6) --, so the ~18
    launches of a call collapse into one graph launch: a single 480x640 image goes from 0.50 to 0.42 ms, a 1080p one from 1.41
    to 1.34 ms (bench.py: ``batch1_latency.graph_replay_wall_ms``); large batches gain nothing (their kernels are long).
    ``det = GraphedDetector(model, example_u8, border, nms_size, num_points)``, then ``idx, score, count, prob = det(images_u8)``
    with ``images_u8`` of the example's shape and dtype on the same device.  The returned tensors are the graph's own output
    buffers: valid until the next call (clone what must outlive it).  The per-call split-f16 status guard does not exist
    inside a replay (nothing on the host looks): the constructor runs ``model.validate_fp16`` on the example instead, and a
    checkpoint that the load-time probes sent to the fp32 kernels is captured on those."""

    def __init__(self, model, example_u8: torch.Tensor, border: int = 15, nms_size: int = 15, num_points: int = 1000):
        if not example_u8.is_cuda or example_u8.dtype != torch.uint8:
            raise ValueError("example_u8 must be a uint8 tensor on the GPU: [B,H,W] gray or [B,H,W,3] RGB")
        dev = example_u8.device
        self._args = (border, nms_size, num_points)
        self._model = model
        self._in = example_u8.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                         # warm-up: weight blob, range probes, kernel attributes, workspaces
            for _ in range(2):
                detect_batch_u8(model, self._in, *self._args)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if model.effective_precision == "fp16":
            h, w = example_u8.shape[1], example_u8.shape[2]
            rgb = self._in if self._in.dim() == 4 else self._in[..., None].expand(-1, -1, -1, 3)
            x = torch.zeros((rgb.shape[0], 3) + arch.padded_hw(h, w)[:2], dtype=torch.float32, device=dev)
            _, _, top, left = arch.padded_hw(h, w)
            x[:, :, top:top + h, left:left + w] = rgb.permute(0, 3, 1, 2).float() / 255.0
            model.validate_fp16(x)                            # raises if the split path leaves its range on the example
        self._graph = torch.cuda.CUDAGraph()
        # thread_local: only this thread must keep to capturable calls (it does: no entry point of the library synchronises or
        # queries); another thread polling an event -- torch's NCCL watchdog does -- would invalidate a "global" capture
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self._out = detect_batch_u8(model, self._in, *self._args)

    def __call__(self, images_u8: torch.Tensor):
        if images_u8.shape != self._in.shape or images_u8.dtype != torch.uint8 or images_u8.device != self._in.device:
            raise ValueError(f"GraphedDetector was captured for uint8 {tuple(self._in.shape)} on {self._in.device}")
        self._in.copy_(images_u8, non_blocking=True)
        self._graph.replay()
        return self._out


def pad_batch(images_rgb_norm: np.ndarray) -> torch.Tensor:
    """[B,H,W,3] float in [0,1] -> padded [B,3,Hp,Wp] float32 CPU tensor (make_shape_even +
    mod_padding_symmetric, test_utils.py:16-32; torch.tensor(...).permute, train_utils.py:426-428)."""
    out = [T.mod_padding_symmetric(T.make_shape_even(im), factor=64) for im in images_rgb_norm]
    return torch.from_numpy(np.stack(out).astype(np.float32)).permute(0, 3, 1, 2).contiguous()


def pad_image_on_device(image_rgb_norm: np.ndarray, device) -> torch.Tensor:
    """[H,W,3] float image in [0,1] (any float dtype; the reference's callers hold float64) -> the padded [1,3,Hp,Wp] float32 batch
    ON THE GPU, bit-identical to ``pad_batch(image[None]).to(device)``: the image is uploaded as it is and cast (round to nearest
    even, like ``astype``), transposed and placed into a zeroed padded tensor by a few device kernels.  ``pad_batch`` does the same
    with four NumPy copies of a 50 MB array per 1080p image: 27 of the 28 ms of an ``extract_detections`` call (round 6,
    tools/caller_latency.py) against 1.4 ms of GPU work."""
    img = np.ascontiguousarray(image_rgb_norm)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"expected an [H,W,3] image, got {img.shape}")
    if img.dtype not in (np.float64, np.float32, np.float16):
        img = img.astype(np.float64)            # (the reference's torch.tensor(..., dtype=float32) accepts any numeric array)
    h, w = img.shape[:2]
    hp, wp, top, left = arch.padded_hw(h, w)
    t = torch.from_numpy(img).to(device)
    x = torch.zeros((1, 3, hp, wp), dtype=torch.float32, device=t.device)
    x[0, :, top:top + h, left:left + w] = t.permute(2, 0, 1).to(torch.float32)
    return x


@torch.no_grad()
def extract_detections(image_RGB_norm, model, device, cell_size=8, nms_size=15, num_points=25, border_size=15):
    h, w = image_RGB_norm.shape[0], image_RGB_norm.shape[1]
    x = pad_image_on_device(image_RGB_norm, device)
    idx, score, count, prob = detect_batch(model, x, h, w, border_size, nms_size, num_points)
    n = int(count[0])                  # (a device-to-host read: the stream has passed the forward)
    # one image per call, results on the host at once: the split-f16 status block is final here, for free -- a flagged call
    # is repeated on the fp32 kernels before anything is returned (mlp_ma_decoder.py: _guarded_call)
    if getattr(model, "fp16_guard_check", None) is not None and model.fp16_guard_check(synchronize=False):
        idx, score, count, prob = detect_batch(model, x, h, w, border_size, nms_size, num_points)
        n = int(count[0])
    i = idx[0, :n].cpu().numpy().astype(np.int64)
    pts = np.empty((n, 4), dtype=np.float64)
    pts[:, 0], pts[:, 1], pts[:, 2], pts[:, 3] = i % w, i // w, 1.0, score[0, :n].cpu().numpy()
    _, _, top, left = arch.padded_hw(h, w)
    return pts, prob[:, top:top + h, left:left + w].unsqueeze(1)


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ``total`` images: rank r gets [lo, hi)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_SHAPES_CHECKED = set()      # (group, world, B, K, device) whose equal-shard precondition has been verified across the group


def allgather_keypoints(idx: torch.Tensor, score: torch.Tensor, count: torch.Tensor, group=None, force: bool = False,
                        total: Optional[int] = None):
    """Every rank ends up with the keypoints of the whole batch, in rank order.

    ``total`` = the number of images of the whole batch when the ranks hold the UNEQUAL shards of
    :func:`shard_range` (a batch that does not divide by the world size): every rank pads its slab to
    ``ceil(total / world)`` rows (index -1, score 0, count 0), the collective moves equal shapes, and the
    padding rows are dropped from the result.  Without ``total`` the shards must be equal: the FIRST call with a given
    (group, B, K) checks the shapes across the group with a small collective of its own and a host read, so that ranks that
    disagree raise instead of hanging inside RCCL; the verdict is cached and every later call with the same shapes is the
    one slab collective, enqueued without any host synchronisation (a rank that changes its shapes alone afterwards is
    outside this check, as with any collective).
    A single-rank group returns its inputs untouched unless ``force`` is set (then the collective runs
    anyway: the plumbing check of SURVEY.md 8e on a one-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return idx, score, count
    if dist.get_world_size(group) == 1 and not force:
        return idx, score, count
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b, k = idx.shape
    if total is None:
        rows = b
        key = (id(group) if group is not None else 0, world, b, k, str(idx.device))
        if key not in _SHAPES_CHECKED:
            # equal shards are a precondition of all_gather_into_tensor: make a violation an error on every rank (once per
            # shape: this costs a second collective and a device-to-host read, which the steady state must not pay)
            mine = torch.tensor([b, k], dtype=torch.int64, device=idx.device)
            seen = torch.empty(world * 2, dtype=torch.int64, device=idx.device)
            dist.all_gather_into_tensor(seen, mine, group=group)
            seen = seen.view(world, 2)
            if not bool((seen == mine).all()):
                raise ValueError(f"allgather_keypoints: slab shapes differ across ranks ({seen.tolist()}); pass total= for the "
                                 f"unequal shards of shard_range")
            _SHAPES_CHECKED.add(key)
        shard_rows = [b] * world
    else:
        rows = -(-total // world)
        shard_rows = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
        if shard_rows[rank] != b:
            raise ValueError(f"allgather_keypoints: rank {rank} of {world} holds {b} images, shard_range({total}) gives {shard_rows[rank]}")
    # one slab per rank: [rows, K] indices, [rows, K] score bits and [rows] counts, packed as int32 so that the
    # step costs a single latency-bound collective
    slab = torch.empty((rows, 2 * k + 1), dtype=torch.int32, device=idx.device)
    slab[:b, :k] = idx
    slab[:b, k:2 * k] = score.view(torch.int32)
    slab[:b, 2 * k] = count
    if rows > b:
        slab[b:, :k] = -1
        slab[b:, k:] = 0
    out = torch.empty((world * rows, 2 * k + 1), dtype=torch.int32, device=idx.device)
    dist.all_gather_into_tensor(out, slab, group=group)
    if any(r != rows for r in shard_rows):                      # drop the padding rows of the short shards
        keep = torch.cat([torch.arange(r * rows, r * rows + shard_rows[r], device=idx.device) for r in range(world)])
        out = out.index_select(0, keep)
    return out[:, :k].contiguous(), out[:, k:2 * k].contiguous().view(torch.float32), out[:, 2 * k].contiguous()
