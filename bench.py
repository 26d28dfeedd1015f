#!/usr/bin/env python3
"""Benchmark of the BALF keypoint-detection hot path on MI355X.

One step = one pass of the hot path (detector forward -> score map -> crop/border/window-max NMS ->
exact top-K -> all-gather of keypoint slabs when N > 1) over one per-GPU batch of synthetic 1080p
grayscale images already resident in HBM (gray replicated to 3 channels, /255, padded to 1088x1920:
SURVEY.md F4/F5).  Workload = BASELINE.json configs[3] divided over the node: 32 images per GPU
(256 over 8 GPUs), top-2000, border 15, nms 15; weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp16|fp32]

--precision fp16 (default) = f16 MFMA with split (hi+lo) operands, three products per tile, fp32
accumulate/LayerNorm/GELU/softmax/NMS: score map within 5e-6 of the reference (north-star bar 1e-4).
--precision fp32 = exact fp32 MFMA.  The other precision is also run (untimed headline, timed on its own)
and reported under "other_precision".
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = images/s over all GPUs; `keypoints_per_s` rides along.
`roofline` is for the dominant kernel (per-kernel device time from hipEvent pairs on the launch
stream over the timed steps: balf_profile_begin/end in include/balf_hip.h); `cpu_baseline` times the
CPU oracle (a port of the reference path, oracle/) on a bounded sample on rank 0 at N = 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from balf_amd import arch, ops, pipeline  # noqa: E402
from balf_amd.model import get_model  # noqa: E402
from balf_amd.utils import synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, dense f32 MFMA
PEAK_FP16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA
PEAK_HBM_GBS = 8000.0


def stage_macs_per_pixel(s: int):
    """Algorithmic MACs per stage-resolution pixel of the two branch kernels:
    grid kernel = u half of dense1 + grid gMLP; block kernel = conv0 + v half + block gMLP + dense2 +
    RCAB convs.  Recomputed work (x0 and LN in the grid kernel) is not counted."""
    c = [32, 64, 128, 256][s]
    cin = [3, 32, 64, 128][s]
    grid = c * c + 2 * c * c + 64 * c + c * c
    block = cin * c + c * c + 2 * c * c + 64 * c + c * c + 2 * c * c + c * c + c * c
    return grid, block


def kernel_bytes_per_launch(name: str, mb: int, hp: int, wp: int) -> float:
    """Algorithmic HBM bytes per launch of a branch kernel: what it must read and write once
    (DESIGN.md 4.1 table): grid: stage input + u'; block: stage input + u' + t + r."""
    if not (name.startswith("stage") and "branch" in name):
        return 0.0
    s = int(name[5]) - 1
    c = [32, 64, 128, 256][s]
    cin = [3, 32, 64, 128][s]
    px = mb * (hp >> s) * (wp >> s)
    per_px = 4.0 * (cin + c) if "grid" in name else 4.0 * (cin + c + 2 * c)
    return per_px * px


def measured_traffic(name: str, precision: str, mb: int, hp: int, wp: int):
    """HBM bytes per launch of `name` from the committed PMC passes (profiles/r1_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc runs of this script at 8 images per launch, 1088x1920,
    gfx950 corrections applied); None when the profile does not cover this shape."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    if not (os.path.isfile(path) and mb == 8 and (hp, wp) == (1088, 1920) and name.startswith("stage")):
        return None
    s = int(name[5]) - 1
    c, cin = [32, 64, 128, 256][s], [3, 32, 64, 128][s]
    suffix = "16" if precision == "fp16" else ""
    if "branch" in name:
        key = f"stage_branch_kernel{suffix}<{c}, {cin}, {0 if 'grid' in name else 1}>"
    elif "pool" in name:
        key = f"pool_kernel{suffix}<{c}>"
    else:
        return None
    try:
        return json.load(open(path))["kernels"][precision][key]["hbm_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def kernel_flops_per_launch(name: str, mb: int, hp: int, wp: int) -> float:
    if name.startswith("stage") and ("grid_branch" in name or "block_branch" in name):
        s = int(name[5]) - 1
        g, b = stage_macs_per_pixel(s)
        px = mb * (hp >> s) * (wp >> s)
        return 2.0 * (g if "grid" in name else b) * px
    if name == "stage4_head":
        return 2.0 * (256 * 256 + 256 * 65) * mb * (hp // 8) * (wp // 8)
    return 0.0


def cpu_baseline(h, w, k, n_images, state, threads=0):
    """The CPU oracle (port of the reference path) on `n_images` of the same workload.  The path is
    layout/elementwise-bound on the CPU and slows down past a few dozen threads, so the
    thread count is capped (256 threads measured 34 s/image on the GPU box, 8 threads 6.6 s in the build
    container)."""
    from oracle import oracle as O
    from oracle import c_oracle
    torch.set_num_threads(threads if threads > 0 else min(os.cpu_count() or 1, 32))
    imgs = np.stack([synth.gray_to_rgb_norm(synth.synthetic_gray_u8(h, w, i)) for i in range(n_images)])
    t0 = time.perf_counter()
    kp = 0
    t_fwd = t_nms = 0.0
    with torch.no_grad():
        for i in range(n_images):
            ta = time.perf_counter()
            pad = O.mod_padding_symmetric(O.make_shape_even(imgs[i]), 64)
            x = torch.tensor(pad, dtype=torch.float32).permute(2, 0, 1).unsqueeze(0)
            prob = O.detector_forward(state, x)["prob"][0].numpy()
            tb = time.perf_counter()
            top, left = O.crop_offsets(h, w, *prob.shape)
            idx, sc, _ = c_oracle.nms_topk(prob[top:top + h, left:left + w], 15, 15, k)
            t_nms += time.perf_counter() - tb
            t_fwd += tb - ta
            kp += idx.size
    dt = time.perf_counter() - t0
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": n_images / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model,
            "keypoints_per_s": kp / dt,
            "forward_s_per_image": t_fwd / n_images, "nms_topk_s_per_image": t_nms / n_images,
            "sample": f"{n_images} synthetic {w}x{h} gray images, batch 1, oracle forward (torch CPU fp32) + C NMS/top-{k}; "
                      f"{dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="fp16", choices=["fp32", "fp16"])
    ap.add_argument("--other-steps", type=int, default=2, help="timed steps of the other precision (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(cores, 32))")
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--topk", type=int, default=2000)
    ap.add_argument("--cpu-images", type=int, default=2, help="images in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: balf_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)

    h, w, k, b = args.height, args.width, args.topk, args.batch_per_gpu
    hp, wp, top, left = arch.padded_hw(h, w)
    state = synth.synthetic_state_dict(20240)
    model = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    model.load_state_dict(state)
    model.precision = args.precision
    model = model.eval().to(dev)

    # synthetic inputs, resident in HBM before the timed region: this rank's shard of the global batch
    lo = rank * b
    gray = np.stack([synth.synthetic_gray_u8(h, w, lo + i, blur=5 if i % 2 == 0 else 1) for i in range(b)])
    g = torch.from_numpy(gray).to(dev).float().div_(255.0)
    x = torch.zeros((b, 3, hp, wp), dtype=torch.float32, device=dev)
    x[:, :, top:top + h, left:left + w] = g[:, None]
    del g

    def step():
        idx, score, count, _ = pipeline.detect_batch(model, x, h, w, 15, 15, k, precomputed_offsets=(top, left))
        return pipeline.allgather_keypoints(idx, score, count)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_run(steps, warmup):
        for _ in range(warmup):
            o = step()
        fence()
        ops.profile_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = step()
        fence()
        dt_ = time.perf_counter() - t0
        prof_ = ops.profile_end()
        if dist is not None:
            tmax = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_, prof_, o

    mb = max(1, min(b, 16 * 1024 * 1024 // (hp * wp)))          # images per launch (make_plan in det_common.h)

    def summarize(precision, dt_, prof_, steps):
        """images/s + the roofline of the dominant kernel: whichever of the MFMA and HBM roofs it sits closer to."""
        name, (ms, n_launch) = max(prof_.items(), key=lambda kv: kv[1][0])
        avg_ms = ms / n_launch
        flops = kernel_flops_per_launch(name, mb, hp, wp)
        nbytes = kernel_bytes_per_launch(name, mb, hp, wp)
        # f16 path: every algorithmic MAC costs three f16 MFMA products (hi*hi + lo*hi + hi*lo)
        peak_tf = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_FP16_MFMA_TFLOPS / 3.0
        tf = flops / (avg_ms * 1e-3) / 1e12 if flops else 0.0
        gbs = nbytes / (avg_ms * 1e-3) / 1e9 if nbytes else 0.0
        f_m, f_h = tf / peak_tf, gbs / PEAK_HBM_GBS
        roof = {"kernel": name, "avg_launch_ms": avg_ms, "launches": n_launch, "images_per_launch": mb,
                "traffic": measured_traffic(name, precision, mb, hp, wp),
                "mfma": {"achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": f_m,
                         "algorithmic_flop_per_launch": flops},
                "hbm": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_h,
                        "algorithmic_bytes_per_launch": nbytes}}
        if f_m >= f_h:
            roof.update({"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": f_m})
        else:
            roof.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_h})
        fwd_ms = sum(v[0] for kname, v in prof_.items() if kname.startswith("stage")) / steps
        nms_ms = sum(v[0] for kname, v in prof_.items() if not kname.startswith("stage")) / steps
        return {"images_per_s": world * b * steps / dt_, "ms_per_step": dt_ / steps * 1e3, "roofline": roof,
                "forward_device_ms_per_step": fwd_ms, "nms_topk_device_ms_per_step": nms_ms,
                "forward_tflops": b * hp * wp * arch.FLOP_PER_PADDED_PIXEL / (fwd_ms * 1e-3) / 1e12,
                "kernels_ms_per_step": {kname: v[0] / steps for kname, v in sorted(prof_.items())}}

    dt, prof, out = timed_run(args.steps, args.warmup)
    counts = out[2]
    kp_per_image = float(counts.float().mean().item())
    head = summarize(args.precision, dt, prof, args.steps)

    other = None
    other_prec = "fp32" if args.precision == "fp16" else "fp16"
    if args.other_steps > 0:
        model.precision = other_prec
        dt2, prof2, _ = timed_run(args.other_steps, 1)
        other = summarize(other_prec, dt2, prof2, args.other_steps)
        other["precision"] = other_prec
        model.precision = args.precision

    if rank == 0:
        ips = head["images_per_s"]
        nms_bytes = b * (4.0 * h * w + 8.0 * k + 4.0)
        nms_ms = head["nms_topk_device_ms_per_step"]
        dtype = ("f16 MFMA, split hi+lo operands (3 products), f32 accumulate/LN/GELU/softmax/NMS"
                 if args.precision == "fp16" else "f32")
        res = {
            "metric": "images/sec + keypoints/sec on 1080p gray (detector forward + NMS + top-K)",
            "value": ips, "unit": "images/s", "keypoints_per_s": ips * kp_per_image,
            "keypoints_per_image": kp_per_image,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{b} images/GPU x {world} GPU, {w}x{h} gray -> [B,3,{hp},{wp}] fp32, top-{k}, "
                                   f"border 15, nms 15 (BASELINE configs[3]/[4] shard)",
                       "global_batch": b * world, "parallelism": f"dp{world}", "precision": args.precision,
                       "collective": "all_gather of [B,2K+1] int32 keypoint slabs" if world > 1 else "none"},
            "roofline": head["roofline"],
            "forward_device_ms_per_step": head["forward_device_ms_per_step"],
            "nms_topk_device_ms_per_step": nms_ms,
            "forward_tflops": head["forward_tflops"],
            "nms_topk_roofline": {"bound": "hbm", "achieved": nms_bytes / (nms_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": nms_bytes / (nms_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "kernels_ms_per_step": head["kernels_ms_per_step"],
            "other_precision": other,
        }
        if world == 1 and args.cpu_images > 0:
            res["cpu_baseline"] = cpu_baseline(h, w, k, args.cpu_images, state, args.cpu_threads)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
