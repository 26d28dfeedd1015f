"""Thin torch-tensor wrappers over the C ABI (device memory and streams are PyTorch's; the compute
is the HIP library's).  Every function requires CUDA tensors and raises otherwise."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib
from ._lib import BalfHipError, check, current_stream_ptr, lib, require_gpu_tensor

_workspaces: Dict[Tuple[str, int, int], torch.Tensor] = {}      # insertion order = least recently used first
_MAX_STREAMS_PER_TAG = 2        # the forward workspace of 8 x 1088x1920 is ~7 GB: a process that keeps creating streams
                                # must not pin one per stream it ever used


def _workspace(tag: str, device, nbytes: int) -> torch.Tensor:
    """Caller-owned scratch (the library never allocates), cached per (purpose, device, STREAM) and grown on demand.
    Kernels of one stream run in order, so one buffer per stream is race-free; two streams (or two models driven from
    two streams) get two buffers.  A buffer that is replaced by a larger one -- or evicted: at most _MAX_STREAMS_PER_TAG
    streams per (purpose, device) keep theirs, least recently used first out -- is handed back to the caching allocator,
    which re-issues it in the order of the stream it was allocated and last used on.  (A raw stream handle may be
    recycled for a new stream after its owner is destroyed; the entry it then finds was last used on the destroyed
    stream, whose work the runtime completes before the handle is reused.)"""
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    key = (tag, dev_index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = None                                            # drop the smaller buffer before asking for the larger one
        with torch.cuda.device(device):
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        same = [k for k in _workspaces if k[0] == tag and k[1] == dev_index]
        for k in same[:max(0, len(same) - (_MAX_STREAMS_PER_TAG - 1))]:
            del _workspaces[k]
    _workspaces[key] = ws                                    # (re)inserted last: most recently used
    return ws


def release_workspaces() -> None:
    _workspaces.clear()


def window_nms(score: torch.Tensor, border: int, nms_size: int) -> torch.Tensor:
    """[B,H,W] fp32 -> dense apply_nms(remove_borders(score, border), nms_size)
    (/root/reference/balf/utils/test_utils.py:34-54)."""
    require_gpu_tensor(score, "score")
    if score.dtype != torch.float32 or score.dim() != 3:
        raise BalfHipError("score must be a [B,H,W] float32 tensor")
    out = torch.empty_like(score)
    b, h, w = score.shape
    with torch.cuda.device(score.device):
        check(lib().balf_window_nms(score.data_ptr(), b, h, w, int(border), int(nms_size), out.data_ptr(),
                                    current_stream_ptr(score.device)), "balf_window_nms")
    return out


def nms_topk(prob: torch.Tensor, crop_y: int, crop_x: int, h: int, w: int, border: int, nms_size: int,
             k: int, threshold: float = -1.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """[B,Hp,Wp] fp32 score maps -> (idx [B,K] int32, score [B,K] fp32, count [B] int32); see
    balf_nms_topk in include/balf_hip.h.  ``k > h*w`` raises IndexError like the reference
    (/root/reference/balf/utils/test_utils.py:83).  ``threshold > 0`` selects by that value instead of the K-th
    largest score (balf_nms_threshold; ``threshold != -1`` of find_index_higher_scores, test_utils.py:91-95)."""
    require_gpu_tensor(prob, "prob")
    if prob.dtype != torch.float32 or prob.dim() != 3:
        raise BalfHipError("prob must be a [B,Hp,Wp] float32 tensor")
    if k > h * w:
        raise IndexError(f"index {k - 1} is out of bounds for axis 0 with size {h * w}")
    b, hp, wp = prob.shape
    dev = prob.device
    idx = torch.empty((b, k), dtype=torch.int32, device=dev)
    score = torch.empty((b, k), dtype=torch.float32, device=dev)
    count = torch.empty((b,), dtype=torch.int32, device=dev)
    nbytes = lib().balf_nms_topk_workspace_bytes(b, h, w, k)
    ws = _workspace("nms", dev, nbytes)
    with torch.cuda.device(dev):
        if threshold == -1:
            check(lib().balf_nms_topk(prob.data_ptr(), b, hp, wp, int(crop_y), int(crop_x), int(h), int(w),
                                      int(border), int(nms_size), int(k), idx.data_ptr(), score.data_ptr(),
                                      count.data_ptr(), ws.data_ptr(), ws.numel(), current_stream_ptr(dev)),
                  "balf_nms_topk")
        else:
            check(lib().balf_nms_threshold(prob.data_ptr(), b, hp, wp, int(crop_y), int(crop_x), int(h), int(w),
                                           int(border), int(nms_size), float(threshold), int(k), idx.data_ptr(),
                                           score.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                                           current_stream_ptr(dev)), "balf_nms_threshold")
    return idx, score, count


def greedy_nms(prob: torch.Tensor, crop_y: int, crop_x: int, h: int, w: int, border: int, conf_thresh: float,
               dist_thresh: int, k: int, subpixel_patch: int = 0):
    """Greedy NMS of the demo path (balf_greedy_nms in include/balf_hip.h).  Returns
    (idx [B,K] int32, score [B,K], xy [B,K,2] or None, count [B], total [B])."""
    require_gpu_tensor(prob, "prob")
    if prob.dtype != torch.float32 or prob.dim() != 3:
        raise BalfHipError("prob must be a [B,Hp,Wp] float32 tensor")
    b, hp, wp = prob.shape
    dev = prob.device
    idx = torch.empty((b, k), dtype=torch.int32, device=dev)
    score = torch.empty((b, k), dtype=torch.float32, device=dev)
    xy = torch.empty((b, k, 2), dtype=torch.float32, device=dev) if subpixel_patch > 0 else None
    count = torch.empty((b,), dtype=torch.int32, device=dev)
    total = torch.empty((b,), dtype=torch.int32, device=dev)
    ws = _workspace("greedy", dev, lib().balf_greedy_nms_workspace_bytes(b, h, w, k))
    with torch.cuda.device(dev):
        check(lib().balf_greedy_nms(prob.data_ptr(), b, hp, wp, int(crop_y), int(crop_x), int(h), int(w), int(border),
                                    float(conf_thresh), int(dist_thresh), int(k), int(subpixel_patch), idx.data_ptr(),
                                    score.data_ptr(), xy.data_ptr() if xy is not None else None, count.data_ptr(),
                                    total.data_ptr(), ws.data_ptr(), ws.numel(), current_stream_ptr(dev)),
              "balf_greedy_nms")
    return idx, score, xy, count, total


def profile_begin() -> None:
    check(lib().balf_profile_begin(), "balf_profile_begin")


def profile_end():
    """-> {slot name: (total device ms, launches)} for the launches since profile_begin()."""
    l = lib()
    n = l.balf_profile_num_slots()
    ms = (C.c_float * n)()
    cnt = (C.c_int * n)()
    check(l.balf_profile_end(ms, cnt), "balf_profile_end")
    return {l.balf_profile_slot_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n) if cnt[i]}


def extract_patches(gray_u8: torch.Tensor, xy: torch.Tensor, scale: float) -> torch.Tensor:
    """uint8 gray image [H,W] + keypoints [N,2] (x, y) -> patches [N,1,32,32] fp32 in [0,1]: what
    ``K.feature.extract_patches_from_pyramid(gray/255, laf_from_center_scale_ori(kp, scale, 0), PS=32)`` returns in
    /root/reference/demo/demo_match.py:62-70 (balf_extract_patches in include/balf_hip.h)."""
    require_gpu_tensor(gray_u8, "gray_u8")
    if not xy.is_cuda:
        raise BalfHipError("xy must live on the GPU")
    if gray_u8.dtype != torch.uint8 or gray_u8.dim() != 2:
        raise BalfHipError("gray_u8 must be a [H,W] uint8 tensor")
    if xy.dim() != 2 or xy.shape[1] != 2:
        raise BalfHipError("xy must be [N,2]")
    xy = xy.contiguous().float()
    n = xy.shape[0]
    h, w = gray_u8.shape
    dev = gray_u8.device
    out = torch.empty((n, 1, 32, 32), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    nbytes = lib().balf_extract_patches_workspace_bytes(h, w, float(scale))
    ws = _workspace("patches", dev, nbytes)
    with torch.cuda.device(dev):
        check(lib().balf_extract_patches(gray_u8.data_ptr(), h, w, xy.data_ptr(), n, float(scale), out.data_ptr(),
                                         ws.data_ptr(), ws.numel(), current_stream_ptr(dev)), "balf_extract_patches")
    return out


def rgb_to_gray_u8(rgb_u8: torch.Tensor) -> torch.Tensor:
    """uint8 RGB [...,3] on the GPU -> uint8 gray [...] with PIL's ``convert('L')`` arithmetic, which is what the
    demo's ``load_im`` feeds the patch extractor (/root/reference/demo/demo_match.py:13-19)."""
    require_gpu_tensor(rgb_u8, "rgb_u8")
    if rgb_u8.dtype != torch.uint8 or rgb_u8.shape[-1] != 3:
        raise BalfHipError("rgb_u8 must be a uint8 tensor with a last dimension of 3")
    out = torch.empty(rgb_u8.shape[:-1], dtype=torch.uint8, device=rgb_u8.device)
    if out.numel():
        with torch.cuda.device(rgb_u8.device):
            check(lib().balf_rgb_to_gray(rgb_u8.data_ptr(), out.numel(), out.data_ptr(), current_stream_ptr(rgb_u8.device)),
                  "balf_rgb_to_gray")
    return out


def extract_patches_batch(gray_u8: torch.Tensor, xy: torch.Tensor, count, scale: float) -> torch.Tensor:
    """Batched :func:`extract_patches`: gray_u8 [B,H,W] uint8, xy [B,K,2], count [B] int32 (or None: all K valid) ->
    patches [B,K,1,32,32]; slots past an image's count are zero patches (balf_extract_patches_batch)."""
    require_gpu_tensor(gray_u8, "gray_u8")
    if gray_u8.dtype != torch.uint8 or gray_u8.dim() != 3:
        raise BalfHipError("gray_u8 must be a [B,H,W] uint8 tensor")
    if xy.dim() != 3 or xy.shape[0] != gray_u8.shape[0] or xy.shape[2] != 2 or not xy.is_cuda:
        raise BalfHipError("xy must be a [B,K,2] GPU tensor")
    xy = xy.contiguous().float()
    b, h, w = gray_u8.shape
    k = xy.shape[1]
    dev = gray_u8.device
    out = torch.empty((b, k, 1, 32, 32), dtype=torch.float32, device=dev)
    if k == 0:
        return out
    if count is not None:
        count = count.to(device=dev, dtype=torch.int32).contiguous()
    ws = _workspace("patches", dev, lib().balf_extract_patches_batch_workspace_bytes(b, h, w, float(scale)))
    with torch.cuda.device(dev):
        check(lib().balf_extract_patches_batch(gray_u8.data_ptr(), b, h, w, xy.data_ptr(),
                                               count.data_ptr() if count is not None else None, k, float(scale),
                                               out.data_ptr(), ws.data_ptr(), ws.numel(), current_stream_ptr(dev)),
              "balf_extract_patches_batch")
    return out


def match_smnn(desc1: torch.Tensor, desc2: torch.Tensor, th: float = 0.8) -> Tuple[torch.Tensor, torch.Tensor]:
    """``kornia.feature.match_smnn(desc1, desc2, th)`` (/root/reference/demo/demo_match.py:104-110): returns
    (dists [M,1] fp32, idxs [M,2] int64), mutual ratio-test matches sorted by the index in ``desc1``."""
    require_gpu_tensor(desc1, "desc1")
    require_gpu_tensor(desc2, "desc2")
    if desc1.dim() != 2 or desc2.dim() != 2 or desc1.shape[1] != 128 or desc2.shape[1] != 128:
        raise BalfHipError("descriptors must be [N,128]")
    desc1, desc2 = desc1.float().contiguous(), desc2.float().contiguous()
    n1, n2 = desc1.shape[0], desc2.shape[0]
    dev = desc1.device
    if n1 == 0 or n2 == 0:
        return (torch.zeros((0, 1), dtype=torch.float32, device=dev), torch.zeros((0, 2), dtype=torch.int64, device=dev))
    cap = min(n1, n2)
    idx = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((cap,), dtype=torch.float32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = _workspace("match", dev, lib().balf_match_smnn_workspace_bytes(n1, n2))
    with torch.cuda.device(dev):
        check(lib().balf_match_smnn(desc1.data_ptr(), n1, desc2.data_ptr(), n2, float(th), idx.data_ptr(),
                                    dist.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                                    current_stream_ptr(dev)), "balf_match_smnn")
    m = int(count.item())
    return dist[:m].view(-1, 1), idx[:m].long()


def match_smnn_batch(desc1: torch.Tensor, n1: torch.Tensor, desc2: torch.Tensor, n2: torch.Tensor, th: float = 0.8):
    """``pairs`` independent :func:`match_smnn` problems in three launches: desc1 [P,K1,128] / desc2 [P,K2,128] with
    n1 / n2 [P] valid rows each -> (dist [P,cap] fp32, idx [P,cap,2] int32 (-1 padded), count [P] int32), cap =
    min(K1, K2); nothing is read back to the host (balf_match_smnn_batch)."""
    require_gpu_tensor(desc1, "desc1")
    require_gpu_tensor(desc2, "desc2")
    if desc1.dim() != 3 or desc2.dim() != 3 or desc1.shape[2] != 128 or desc2.shape[2] != 128 or desc1.shape[0] != desc2.shape[0]:
        raise BalfHipError("descriptors must be [P,K,128] with the same number of pairs")
    desc1, desc2 = desc1.float(), desc2.float()
    p, k1, k2 = desc1.shape[0], desc1.shape[1], desc2.shape[1]
    dev = desc1.device
    cap = min(k1, k2)
    idx = torch.empty((p, cap, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((p, cap), dtype=torch.float32, device=dev)
    count = torch.empty((p,), dtype=torch.int32, device=dev)
    if p == 0 or cap == 0:
        return dist, idx, count.zero_()
    n1 = n1.to(device=dev, dtype=torch.int32).contiguous()
    n2 = n2.to(device=dev, dtype=torch.int32).contiguous()
    ws = _workspace("match", dev, lib().balf_match_smnn_batch_workspace_bytes(p, k1, k2))
    with torch.cuda.device(dev):
        check(lib().balf_match_smnn_batch(desc1.data_ptr(), k1, n1.data_ptr(), desc2.data_ptr(), k2, n2.data_ptr(), p,
                                          float(th), idx.data_ptr(), dist.data_ptr(), count.data_ptr(), ws.data_ptr(),
                                          ws.numel(), current_stream_ptr(dev)), "balf_match_smnn_batch")
    return dist, idx, count
