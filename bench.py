#!/usr/bin/env python3
"""Benchmark of the BALF keypoint-detection hot path on MI355X.

One step = one pass of the hot path (detector forward -> score map -> crop/border/window-max NMS ->
exact top-K -> all-gather of keypoint slabs) over one per-GPU batch of synthetic 1080p grayscale images
already resident in HBM (gray replicated to 3 channels, /255, padded to 1088x1920: SURVEY.md F4/F5).
Workload = BASELINE.json configs[3] divided over the node: 32 images per GPU (256 over 8 GPUs), top-2000,
border 15, nms 15; weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--precision fp16 (default, and the module default) = f16 MFMA with split (hi+lo) operands, three products per
MAC, fp32 accumulate/LayerNorm/GELU/softmax/NMS: score map within 1e-4 of the CPU reference (the north-star bar;
measured 4e-6 ... 6e-6 -- `index_match.prob_max_abs_err` in the line says what this run measured).
--precision fp32 = exact fp32 MFMA.  The other precision is timed too and reported under "other_precision".

Rank 0 prints ONE JSON line.  `value` = images/s over all GPUs; `keypoints_per_s` rides along.
  roofline      the dominant kernel (per-kernel device time from hipEvent pairs on the launch stream over the timed
                steps: balf_profile_begin/end in include/balf_hip.h) against the HBM roof (8 TB/s) and the dense f16 /
                f32 MFMA peak of MI355X_MICROARCH.md; `traffic` = measured HBM bytes per launch and `issue` = the share
                of the SIMDs' issue cycles its vector + matrix instructions need, both from the committed rocprofv3
                --pmc passes of this script (profiles/rN_pmc.json, newest round); "bound" says which limit the kernel sits at.
  index_match   BASELINE's "NMS index match vs CPU ref": the images of the CPU sample against the oracle.
  cpu_baseline  the CPU oracle (a port of the reference path, oracle/) on a bounded sample, rank 0, N = 1.
  sustained     the same step back to back for >= 10 s with the clock / package power medians (the 20-step figure is 0.7 s).
  host_fed      the reference's real calling pattern: uint8 gray images in (pinned) HOST memory -> H2D on a copy stream,
                double-buffered -> detect_batch_u8 (logits on) -> [B,2K+1] keypoint slabs back in host memory; images/s and
                the ratio to the resident figure.  Never `value`.
  other_configs the other BASELINE configurations that fit one GPU, a few steps each, every entry with the roofline of ITS
                dominant kernel (hipEvents of that run; PMC traffic only for the profiled shape) and a CPU baseline on two
                of its images; `cpu_baseline.batched` is SURVEY 8(d)'s batch-min(B, 8) leg of the oracle at VGA.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _import_product():
    """torch, NumPy and the product package, imported only by a process that is going to be a rank: the launcher parent
    of `--gpus N` (launch_ranks) must not load anything that could initialise the GPU (loading libbalf_hip.so registers
    its code objects with the HIP runtime)."""
    global np, torch, arch, ops, pipeline, get_model, synth
    import numpy as np
    import torch
    from balf_amd import arch, ops, pipeline
    from balf_amd.model import get_model
    from balf_amd.utils import synth


PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, dense f32 MFMA
PEAK_FP16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA (the hardware's peak: the split path spends 3 products per MAC)
PEAK_HBM_GBS = 8000.0
def _newest_pmc_profile():
    import glob
    import re
    found = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")):
        m = re.fullmatch(r"r(\d+)_pmc\.json", os.path.basename(f))      # (a stray file that merely matches the glob is skipped)
        if m:
            found.append((int(m.group(1)), f))
    return max(found)[1] if found else os.path.join(ROOT, "profiles", "r3_pmc.json")


PMC_PROFILE = _newest_pmc_profile()
PMC_DEFAULT_SHAPE = {"images_per_launch": 16, "hp": 1088, "wp": 1920}   # profiles written before the shape was recorded in them
C_STAGE = [32, 64, 128, 256]
CIN_STAGE = [3, 32, 64, 128]


def stage_macs_per_pixel(s: int):
    """Algorithmic MACs per stage-resolution pixel of the two branch kernels:
    grid kernel = u half of dense1 + grid gMLP; block kernel = conv0 + v half + block gMLP + dense2 +
    RCAB convs.  Recomputed work (x0 and LN in the grid kernel) is not counted."""
    c, cin = C_STAGE[s], CIN_STAGE[s]
    grid = c * c + 2 * c * c + 64 * c + c * c
    block = cin * c + c * c + 2 * c * c + 64 * c + c * c + 2 * c * c + c * c + c * c
    return grid, block


def kernel_bytes_per_launch(name: str, mb: int, hp: int, wp: int, in_bytes_per_px: float = 12.0,
                            precision_is_f16: bool = True) -> float:
    """Algorithmic HBM bytes per launch of the split-f16 schedule: what the kernel must read and write once.
    grid: stage input + u'; block: stage input + u' + x1 (stages 1-3: the tail kernel recomputes the RCAB branch) or
    + t + r (stage 4); pool slot = tail kernel: stage input + x1 (4 pixels) + pooled output; head: t + r + prob."""
    if not name.startswith("stage"):
        return 0.0
    s = int(name[5]) - 1
    c, cin = C_STAGE[s], CIN_STAGE[s]
    px = mb * (hp >> s) * (wp >> s)
    x_in = in_bytes_per_px if s == 0 else 4.0 * cin
    fused = s < 3 and precision_is_f16
    if "grid_branch" in name:
        return (x_in + 4.0 * c) * px
    if "block_branch" in name:
        return (x_in + 4.0 * c + (4.0 if fused else 8.0) * c) * px
    if "pool" in name:
        return ((x_in + 4.0 * c if fused else 8.0 * c) + 4.0 * c / 4.0) * px
    if "head" in name:
        return (8.0 * c + 64.0 * 4.0) * px
    return 0.0


def kernel_flops_per_launch(name: str, mb: int, hp: int, wp: int) -> float:
    if name.startswith("stage") and ("grid_branch" in name or "block_branch" in name):
        s = int(name[5]) - 1
        g, b = stage_macs_per_pixel(s)
        px = mb * (hp >> s) * (wp >> s)
        return 2.0 * (g if "grid" in name else b) * px
    if name == "stage4_head":
        return 2.0 * (256 * 256 + 256 * 65) * mb * (hp // 8) * (wp // 8)
    return 0.0


def pmc_profile(precision: str, mb: int, hp: int, wp: int):
    """The committed PMC passes (profiles/rN_pmc.json of the newest round, tools/pmc_json.py): per profile slot, HBM bytes per launch and the
    issue-slot accounting; None when they do not cover this shape."""
    if not os.path.isfile(PMC_PROFILE):
        return None
    try:
        prof = json.load(open(PMC_PROFILE))
        shape = prof.get("shape", PMC_DEFAULT_SHAPE)
        if (mb, hp, wp) != (shape["images_per_launch"], shape["hp"], shape["wp"]):
            return None                          # per-launch bytes and instruction counts of another launch size: not comparable
        return prof["slots"][precision]
    except (KeyError, ValueError):
        return None


def synthetic_batch(h, w, lo, b):
    """uint8 gray images lo .. lo+b-1 of the bench (alternating blur: smooth and noisy score maps)."""
    return np.stack([synth.synthetic_gray_u8(h, w, lo + i, blur=5 if i % 2 == 0 else 1) for i in range(b)])


def cpu_baseline(gray, k, state, threads=0, batch=1):
    """The CPU oracle (port of the reference path) on the images `gray` [n,H,W] uint8 of the same workload.  The path
    is layout/elementwise-bound on the CPU and slows down past a few dozen threads, so the thread count
    is capped (256 threads measured 34 s/image on the GPU box, 8 threads 6.6 s in the build container).
    batch > 1: the forward runs on `batch` images per call (SURVEY 8d's batch-min(B, 8) leg; the reference itself only
    ever calls the model with one image, /root/reference/balf/utils/train_utils.py:428).
    Returns (report, [padded score maps], [(idx raster order, score)])."""
    from oracle import oracle as O
    from oracle import c_oracle
    torch.set_num_threads(threads if threads > 0 else min(os.cpu_count() or 1, 32))
    n_images, h, w = gray.shape
    imgs = np.stack([synth.gray_to_rgb_norm(g) for g in gray])
    t0 = time.perf_counter()
    kp = 0
    t_fwd = t_nms = 0.0
    probs, dets = [], []
    with torch.no_grad():
        for i0 in range(0, n_images, batch):
            ta = time.perf_counter()
            xs = []
            for i in range(i0, min(i0 + batch, n_images)):
                pad = O.mod_padding_symmetric(O.make_shape_even(imgs[i]), 64)
                xs.append(torch.tensor(pad, dtype=torch.float32).permute(2, 0, 1))
            pb = O.detector_forward(state, torch.stack(xs))["prob"].numpy()
            tb = time.perf_counter()
            for prob in pb:
                top, left = O.crop_offsets(h, w, *prob.shape)
                idx, sc, _ = c_oracle.nms_topk(np.ascontiguousarray(prob[top:top + h, left:left + w]), 15, 15, k)
                kp += idx.size
                probs.append(prob)
                dets.append((idx, sc))
            t_nms += time.perf_counter() - tb
            t_fwd += tb - ta
    dt = time.perf_counter() - t0
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    rep = {"value": n_images / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "host_cpus": os.cpu_count(), "cpu_model": cpu_model,
           "keypoints_per_s": kp / dt,
           "forward_s_per_image": t_fwd / n_images, "nms_topk_s_per_image": t_nms / n_images,
           "batch": batch,
           "sample": f"{n_images} synthetic {w}x{h} gray images (the first of the GPU batch), batch {batch}, oracle forward "
                     f"(torch CPU fp32) + C NMS/top-{k}; {dt:.1f} s"}
    return rep, probs, dets


def index_match(gpu, cpu_probs, cpu_dets, h, w, k, top, left):
    """BASELINE's "NMS index match vs CPU ref" on the sampled images (reference pipeline:
    /root/reference/balf/utils/train_utils.py:416-454).  `identical_input`: the C oracle's NMS/top-K on the very score
    map the GPU produced gives the GPU's indices and score bits; `end_to_end_overlap`: GPU score map -> GPU NMS against
    oracle score map -> oracle NMS (fraction of the K indices shared, worst image); `prob_max_abs_err`: the two score maps."""
    from oracle import oracle as O
    from oracle import c_oracle
    idx, score, count, prob = gpu
    ident, overlap, err = True, 1.0, 0.0
    for i, (cp, (ci, _)) in enumerate(zip(cpu_probs, cpu_dets)):
        gp = prob[i].cpu().numpy()
        err = max(err, float(np.abs(gp - cp).max()))
        ri, rs, _ = c_oracle.nms_topk(np.ascontiguousarray(gp[top:top + h, left:left + w]), 15, 15, k)
        ri, rs = O.canonical_order(ri.astype(np.int64), rs)
        n = int(count[i])
        gi, gs = idx[i, :n].cpu().numpy(), score[i, :n].cpu().numpy()
        ident = ident and n == ri.size and np.array_equal(gi, ri.astype(np.int32)) and \
            np.array_equal(gs.view(np.uint32), rs.view(np.uint32))
        overlap = min(overlap, len(set(gi.tolist()) & set(ci.tolist())) / float(max(1, ci.size)))
    return {"images": len(cpu_probs), "identical_input": bool(ident), "end_to_end_overlap": overlap,
            "prob_max_abs_err": err, "tolerance": 1e-4}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_environment(env):
    """What every rank of this bench needs in its environment, set in ONE place (launch_ranks for the ranks it starts,
    main() for a rank some other launcher started, tests/test_rccl_gpu.py for its child): RCCL shares device buffers between
    the ranks of a node through IPC handles, and the hosts of this pool only support the dmabuf kind -- without
    HSA_ENABLE_IPC_MODE_LEGACY=0 a multi-rank group fails with `hipIpcGetMemHandle: invalid argument`.  A single-rank
    group exchanges no handles and comes up either way (measured in round 3), which is why N = 1 never showed it."""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return env


RANK_GRACE_S = float(os.environ.get("BALF_BENCH_GRACE_S", "20"))   # between terminate() and kill() of the ranks that survive a failed one
LAUNCH_DEADLINE_S = float(os.environ.get("BALF_BENCH_DEADLINE_S", "1500"))     # the whole N-rank run


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N fresh ranks of this script (one per GPU) and relay
    rank 0's JSON line.  This parent never touches the GPU (no torch.cuda call, no HIP call): a process that has
    initialised the GPU must not be replaced or forked on this pool, so the ranks are plain child processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set.
    Returns the exit status: 0 only if every rank exited 0 and rank 0 printed its line."""
    import subprocess
    env0 = rank_environment(dict(os.environ))
    env0.setdefault("MASTER_PORT", str(_free_port()))
    env0["WORLD_SIZE"] = env0["LOCAL_WORLD_SIZE"] = str(n)
    env0["BALF_BENCH_LAUNCHED"] = "1"
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    # rank 0's stdout carries the one JSON line: a thread drains it while this one watches the ranks; the first rank that
    # fails takes the others down with it (a rank waiting in a rendezvous for a dead peer would otherwise sit out its timeout)
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    status, live = 0, dict(enumerate(procs))
    t_start = time.monotonic()
    t_term = None                                    # when the survivors were told to stop
    while live:
        for r, p in list(live.items()):
            rc = p.poll()
            if rc is None:
                continue
            del live[r]
            if rc != 0:
                print(f"[bench] rank {r} exited with status {rc}", file=sys.stderr)
                if status == 0:
                    status = rc if rc > 0 else 1
        now = time.monotonic()
        if status == 0 and now - t_start > LAUNCH_DEADLINE_S:
            print(f"[bench] the ranks did not finish within {LAUNCH_DEADLINE_S:.0f} s", file=sys.stderr)
            status = 1
        if status != 0 and live:
            if t_term is None:
                t_term = now
                for q in live.values():
                    q.terminate()             # these exact children, nothing matched by pattern
            elif now - t_term > RANK_GRACE_S:
                # a rank stuck in an RCCL or driver call may ignore SIGTERM: do not poll forever
                for r, q in live.items():
                    print(f"[bench] rank {r} ignored SIGTERM for {RANK_GRACE_S:.0f} s: killing it", file=sys.stderr)
                    q.kill()
                t_term = now + 3600.0         # (kill() cannot be ignored; wait() below reaps them)
                for q in live.values():
                    try:
                        q.wait(timeout=10.0)
                    except subprocess.TimeoutExpired:
                        pass
                live = {}
        time.sleep(0.05)
    reader.join(timeout=10.0)
    line0 = (buf[0] if buf else b"").decode()
    lines = [ln for ln in line0.splitlines() if ln.strip()]
    if status == 0 and len(lines) != 1:
        print(f"[bench] rank 0 printed {len(lines)} lines instead of one JSON line", file=sys.stderr)
        status = 1
    if status == 0:
        sys.stdout.write(lines[0] + "\n")
        sys.stdout.flush()
    return status


def stub_main(args, json_fd):
    """--stub-step: the launch / rendezvous / collective / report plumbing of this script with the GPU taken out — a gloo
    group on the CPU, fixed fake keypoint slabs as the step's result.  It exists for tests/test_bench_launcher.py (the
    N > 1 path of bench.py on a box without GPUs) and is marked as such in its line; it measures nothing."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            raise SystemExit("[bench] MASTER_PORT is not set: the ranks of one run must agree on it")
        os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("BALF_BENCH_TEST_FAULT") == "hang-after-failure":
        # (tests/test_bench_launcher.py) rank 0 fails at once, the others ignore SIGTERM: the launcher must kill them
        if rank == 0:
            raise SystemExit(3)
        import signal
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(3600)
    if os.environ.get("BALF_BENCH_TEST_FAULT") == "hang-all":
        time.sleep(3600)
    b, k = args.batch_per_gpu, args.topk
    total, lo = None, rank * b
    if args.global_batch:                      # the unequal shards of pipeline.shard_range, padded inside the collective
        total = args.global_batch
        lo, hi = pipeline.shard_range(total, rank, world)
        b = hi - lo
    g = torch.Generator().manual_seed(77 + rank)
    idx = torch.randint(0, 1 << 20, (b, k), generator=g, dtype=torch.int32)
    score, count = torch.rand((b, k), generator=g), torch.full((b,), k, dtype=torch.int32)

    def step():
        return pipeline.allgather_keypoints(idx, score, count, force=True, total=total)
    for _ in range(args.warmup):
        out = step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, -dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, dt_min = float(t[0]), -float(t[1])
    n_all = total if total is not None else world * b
    assert out[0].shape == (n_all, k) and torch.equal(out[0][lo:lo + b], idx)
    # what a real rank binds: cuda:LOCAL_RANK (main() does torch.cuda.set_device(local_rank)); the stub has no GPU and only reports it
    rank_devices = [None] * world
    dist.all_gather_object(rank_devices, {"rank": rank, "local_rank_env": int(os.environ.get("LOCAL_RANK", "0")),
                                          "would_bind": f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}", "images": [lo, lo + b]})
    if rank == 0:
        b = n_all / world                      # (mean images per rank, for the line below)
        res = {"metric": "bench.py launcher plumbing (stub step: no detector, no GPU)", "stub": True,
               "value": world * b * args.steps / dt_max, "unit": "images/s", "n_gpus": world,
               "rccl_ranks": dist.get_world_size(), "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "none", "data": "synthetic",
               "per_rank_images_per_s": {"min": b * args.steps / dt_max, "max": b * args.steps / dt_min},
               "config": {"workload": f"stub: {b} x {k} fake keypoints per rank, all-gather only", "global_batch": n_all,
                          "parallelism": f"dp{world}"},
               "launched_by": "bench.py" if os.environ.get("BALF_BENCH_LAUNCHED") else "external launcher",
               "rank_devices": rank_devices,
               "rank_env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}}
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


def single_image_latency(model, dev, h, w, k, calls=60):
    """Median over `calls` single-image detect_batch_u8 calls (uint8 gray image resident in HBM -> forward -> NMS -> top-K),
    each followed by a synchronisation.  wall = host clock around call + sync; device = hipEvents recorded on the launch
    stream right before and after the call (first launch to last kernel's end, launch gaps included); kernels = sum of the
    kernels' own durations (a separate profiled pass).  host_overhead_us = wall - device."""
    import statistics
    img = torch.from_numpy(synthetic_batch(h, w, 0, 1)).to(dev)
    for _ in range(5):
        pipeline.detect_batch_u8(model, img, 15, 15, k)
    torch.cuda.synchronize(dev)
    walls, devs = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(calls):
        t0 = time.perf_counter()
        e0.record()
        out = pipeline.detect_batch_u8(model, img, 15, 15, k)
        e1.record()
        torch.cuda.synchronize(dev)
        walls.append((time.perf_counter() - t0) * 1e3)
        devs.append(e0.elapsed_time(e1))
    ops.profile_begin()
    for _ in range(10):
        pipeline.detect_batch_u8(model, img, 15, 15, k)
    torch.cuda.synchronize(dev)
    prof = ops.profile_end()
    kern = sum(v[0] for v in prof.values()) / 10
    wall, devm = statistics.median(walls), statistics.median(devs)
    # the same call captured ONCE into a hipGraph (torch.cuda.graph) and replayed: the library is stream-ordered end to end (no
    # entry point synchronises or reads back), so the 18 launches of a single-image call collapse into one graph launch
    graph_wall = None
    try:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                pipeline.detect_batch_u8(model, img, 15, 15, k)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        # (thread_local: the NCCL watchdog thread of the single-rank group polls its events meanwhile; in the default "global" mode
        # an event query from ANY thread invalidates a capture)
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            g_out = pipeline.detect_batch_u8(model, img, 15, 15, k)
        gw = []
        for _ in range(calls):
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize(dev)
            gw.append((time.perf_counter() - t0) * 1e3)
        if torch.equal(g_out[0], out[0]) and torch.equal(g_out[2], out[2]):
            graph_wall = statistics.median(gw)
        del graph, g_out
    except Exception as e:          # noqa: BLE001 -- a diagnostic: never fail the bench over it
        print(f"[bench] graph replay leg skipped: {type(e).__name__}: {e}", file=sys.stderr)
    # the reference's own calling convention: extract_detections(image_RGB_norm float64 [H,W,3] on the HOST, ...) -> NumPy points
    # (/root/reference/balf/utils/train_utils.py:416-454), upload, padding and the result's device-to-host copies included
    caller_ms = None
    try:
        img64 = synth.gray_to_rgb_norm(synthetic_batch(h, w, 0, 1)[0]).astype(np.float64)
        for _ in range(3):
            pipeline.extract_detections(img64, model, dev, nms_size=15, num_points=k, border_size=15)
        cw = []
        for _ in range(20):
            t0 = time.perf_counter()
            pipeline.extract_detections(img64, model, dev, nms_size=15, num_points=k, border_size=15)
            cw.append((time.perf_counter() - t0) * 1e3)
        caller_ms = statistics.median(cw)
    except Exception as e:          # noqa: BLE001
        print(f"[bench] extract_detections leg skipped: {type(e).__name__}: {e}", file=sys.stderr)
    return {"workload": f"1 x {w}x{h} uint8 gray, top-{k} (detect_batch_u8 + sync)", "calls": calls, "wall_ms": wall,
            "graph_replay_wall_ms": graph_wall, "extract_detections_host_image_wall_ms": caller_ms,
            "device_ms": devm, "kernels_ms": kern, "host_overhead_us": (wall - devm) * 1e3,
            "wall_minus_kernels_us": (wall - kern) * 1e3, "wall_ms_p90": sorted(walls)[int(0.9 * calls)],
            "images_per_s": 1e3 / wall, "launches": sum(v[1] for v in prof.values()) // 10, "keypoints": int(out[2][0])}


def power_state_under_load(step, dev, seconds=2.5):
    """Clock and package power while the step runs back to back (rocm-smi, sampled from a thread): the forward runs AT the
    package power cap of the MI355X (~1380 W of 1400), which holds the shader clock near 2.0 GHz instead of the 2.4 GHz the
    peak figures assume -- the roofline's `peak` is the nominal one, this field says what the silicon sustained."""
    import re
    import subprocess
    import threading
    samples = []

    def sampler():
        for _ in range(3):
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                d = next(iter(json.loads(o[o.index("{"):]).values()))
                clk = re.search(r"(\d+)", d.get("sclk clock speed:", ""))
                pw = next((float(v) for k_, v in d.items() if "Power (W)" in k_), None)
                if clk and pw is not None:
                    samples.append((int(clk.group(1)), pw))
            except Exception:          # noqa: BLE001 -- a diagnostic: never fail the bench over it
                return
            time.sleep(0.3)
    th = threading.Thread(target=sampler, daemon=True)
    t_end = time.perf_counter() + seconds
    for _ in range(4):
        step()
    torch.cuda.synchronize(dev)
    th.start()
    while time.perf_counter() < t_end or th.is_alive():
        for _ in range(4):
            step()
        torch.cuda.synchronize(dev)
        if time.perf_counter() > t_end + 6.0:
            break
    th.join(timeout=1.0)
    if not samples:
        return None
    return {"sclk_mhz": sorted(c for c, _ in samples)[len(samples) // 2], "package_power_w": sorted(p_ for _, p_ in samples)[len(samples) // 2],
            "samples": len(samples), "nominal_sclk_mhz": 2400, "package_power_cap_w": 1400,
            "note": "median of rocm-smi samples while the timed step runs back to back"}


def sustained_run(step, dev, images_per_step, seconds=10.0):
    """The timed step back to back for >= `seconds` (groups of 4 steps, one synchronisation per group), rocm-smi sampled from a
    thread all the way: the 20-step headline is 0.7 s of a cold-ish package; this is the throughput, clock and power the
    silicon sustains."""
    import re
    import subprocess
    import threading
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                d = next(iter(json.loads(o[o.index("{"):]).values()))
                clk = re.search(r"(\d+)", d.get("sclk clock speed:", ""))
                pw = next((float(v) for k_, v in d.items() if "Power (W)" in k_), None)
                if clk and pw is not None:
                    samples.append((int(clk.group(1)), pw))
            except Exception:          # noqa: BLE001 -- a diagnostic: never fail the bench over it
                return
            stop.wait(0.4)
    for _ in range(4):
        step()
    torch.cuda.synchronize(dev)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            step()
        torch.cuda.synchronize(dev)
        n += 4
    dt = time.perf_counter() - t0
    stop.set()
    th.join(timeout=6.0)
    out = {"seconds": dt, "steps": n, "images_per_s": n * images_per_step / dt, "ms_per_step": dt / n * 1e3}
    if samples:
        med = lambda v: sorted(v)[len(v) // 2]          # noqa: E731
        out.update({"sclk_mhz": med([c for c, _ in samples]), "package_power_w": med([p_ for _, p_ in samples]),
                    "smi_samples": len(samples),
                    "joule_per_image": med([p_ for _, p_ in samples]) / (n * images_per_step / dt)})
    return out


def host_fed_run(model, dev, gray_u8, k, steps, resident_step):
    """End to end from host memory (the reference's caller starts from a host array and ends with host points,
    /root/reference/balf/utils/train_utils.py:426-434): uint8 gray batches in PINNED host memory -> H2D on a copy stream into one
    of two device buffers while the previous batch computes -> detect_batch_u8 with the logits ON -> the [B, 2K+1] int32
    keypoint slab back into pinned host memory.  1 byte per pixel crosses PCIe instead of the 12 of the padded fp32 batch."""
    b, h, w = gray_u8.shape
    host_in = [torch.from_numpy(gray_u8).clone().pin_memory() for _ in range(2)]
    dev_in = [torch.empty((b, h, w), dtype=torch.uint8, device=dev) for _ in range(2)]
    host_out = [torch.empty((b, 2 * k + 1), dtype=torch.int32).pin_memory() for _ in range(2)]
    slab = [torch.empty((b, 2 * k + 1), dtype=torch.int32, device=dev) for _ in range(2)]
    copy_s = torch.cuda.Stream(dev)
    main_s = torch.cuda.current_stream(dev)
    ev_in = [torch.cuda.Event() for _ in range(2)]       # H2D of buffer i done
    ev_free = [torch.cuda.Event() for _ in range(2)]     # the forward that read buffer i is done
    ev_out = [torch.cuda.Event() for _ in range(2)]      # slab i is in host memory

    def upload(i):
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(ev_free[i])
            dev_in[i].copy_(host_in[i], non_blocking=True)
            ev_in[i].record(copy_s)

    def compute(i):
        main_s.wait_event(ev_in[i])
        out = model.forward_u8(dev_in[i], want_logits=True)
        ev_free[i].record(main_s)
        _, _, top_, left_ = arch.padded_hw(h, w)
        idx, score, count = ops.nms_topk(out["prob"], top_, left_, h, w, 15, 15, k)
        slab[i][:, :k] = idx
        slab[i][:, k:2 * k] = score.view(torch.int32)
        slab[i][:, 2 * k] = count
        host_out[i].copy_(slab[i], non_blocking=True)
        ev_out[i].record(main_s)
        return idx

    def run(n):
        for i in range(2):
            ev_free[i].record(main_s)
        upload(0)
        copy_s.synchronize()                          # steady state: in a continuous feed this upload ran under the previous batch
        consumed = 0
        t_start[0] = time.perf_counter()
        for s_ in range(n):
            i = s_ & 1
            if s_ + 1 < n:
                upload(i ^ 1)                         # the next batch crosses PCIe while this one computes
            if s_ >= 2:
                ev_out[i].synchronize()               # the host takes slab i (step s_ - 2) before it is overwritten
                consumed += int(host_out[i][0, 2 * k])
            last = compute(i)
        torch.cuda.synchronize(dev)
        return last, consumed

    def resident(n):                                  # the headline step, timed right before and right after (same thermal state)
        resident_step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            resident_step()
        torch.cuda.synchronize(dev)
        return b * n / (time.perf_counter() - t0)
    t_start = [0.0]
    run(3)
    res_before = resident(max(steps // 2, 4))
    last, _ = run(steps)
    dt = time.perf_counter() - t_start[0]
    i_last = (steps - 1) & 1
    ok = bool(torch.equal(host_out[i_last][:, :k], last.cpu())) and int(host_out[i_last][:, 2 * k].min()) > 0
    ips = b * steps / dt
    res_ips = 0.5 * (res_before + resident(max(steps // 2, 4)))
    return {"images_per_s": ips, "ms_per_step": dt / steps * 1e3, "steps": steps, "ratio_to_resident": ips / res_ips,
            "resident_images_per_s_adjacent": res_ips,
            "logits": True, "h2d_bytes_per_step": int(b * h * w), "d2h_bytes_per_step": int(b * (2 * k + 1) * 4),
            "slabs_on_host_match_device": ok,
            "path": "pinned uint8 gray -> H2D (copy stream, 2 buffers) -> balf_forward_u8 (+logits) -> balf_nms_topk -> [B,2K+1] int32 slab -> pinned host"}


def natural_match(model, dev):
    """index_match on the three photographs / poster of tests/golden/natural.npz against the REFERENCE'S OWN outputs recorded there
    (its extract_detections points and score-map samples; tests/golden/make_golden.py): identical-input NMS parity through the C
    oracle, end-to-end overlap with the reference's points, score map error on the recorded samples."""
    from oracle import oracle as O
    from oracle import c_oracle
    from tests.golden import cases
    f = np.load(os.path.join(ROOT, "tests", "golden", "natural.npz"))
    ident, overlap, err, n_img = True, 1.0, 0.0, 0
    for name, (k_, border, nms) in cases.NATURAL_CASES.items():
        im = cases.poster_u8() if name == "poster" else f[name + ".u8"]
        h_, w_ = im.shape[:2]
        _, _, top_, left_ = arch.padded_hw(h_, w_)
        img = torch.from_numpy(np.ascontiguousarray(im)).to(dev)[None]
        idx, score, count, prob = pipeline.detect_batch_u8(model, img, border, nms, k_)
        gp = prob[0].cpu().numpy()
        err = max(err, float(np.abs(gp[::8, ::8] - f[name + ".prob_s8"]).max()))
        ri, rs, _ = c_oracle.nms_topk(np.ascontiguousarray(gp[top_:top_ + h_, left_:left_ + w_]), border, nms, k_)
        ri, rs = O.canonical_order(ri.astype(np.int64), rs)
        n = int(count[0])
        gi = idx[0, :n].cpu().numpy()
        ident = ident and n == ri.size and np.array_equal(gi, ri.astype(np.int32))
        ref = f[name + ".pts"]
        rset = set((ref[:, 1] * w_ + ref[:, 0]).astype(np.int64).tolist())
        overlap = min(overlap, len(rset & set(gi.tolist())) / float(len(rset)))
        n_img += 1
    return {"images": n_img, "identical_input": bool(ident), "end_to_end_overlap_with_reference_points": overlap,
            "prob_max_abs_err_on_recorded_samples": err, "source": "tests/golden/natural.npz (the reference's own outputs)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="fp16", choices=["fp32", "fp16"])
    ap.add_argument("--other-steps", type=int, default=2, help="timed steps of the other precision (0 = skip)")
    ap.add_argument("--other-configs", type=int, default=1, help="also time the other BASELINE configurations (N = 1)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(cores, 32))")
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="images of the WHOLE job instead of --batch-per-gpu; must divide by --gpus (the timed step gathers equal "
                         "slabs; pipeline.allgather_keypoints(total=) pads unequal shards, which only the stub step exercises)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--topk", type=int, default=2000)
    ap.add_argument("--cpu-images", type=int, default=8,
                    help="images in the CPU-baseline / index_match sample, spread over the batch's micro-batches (0 = skip)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="length of the sustained leg (0 = skip)")
    ap.add_argument("--host-fed-steps", type=int, default=20, help="steps of the host-fed leg (0 = skip)")
    ap.add_argument("--no-single-rank-collective", action="store_true",
                    help="at N = 1 skip the single-rank RCCL group (then the step has no collective)")
    ap.add_argument("--allow-diagnostic-build", action="store_true",
                    help="time a library built with a switch of csrc/diag.h (BALF_HIP_LIB=...); its metric field says INVALID")
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)   # launcher test on the CPU (gloo), see stub_main
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the driver's plain `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank_environment(os.environ)               # (before anything loads the HIP runtime)
    _import_product()

    # stdout carries exactly ONE line, the JSON: libraries that print banners there (RCCL prints its version block when
    # the first communicator comes up) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # never report one GPU count under another's name: the scaling run divides by --gpus
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (plain "
                         f"`python bench.py --gpus {args.gpus}` does it itself) or pass --gpus {world}")
    if args.global_batch and not args.stub_step:
        if args.global_batch % world:
            raise SystemExit(f"[bench] --global-batch {args.global_batch} does not divide by --gpus {world}: the measured step shards the "
                             f"batch equally (weak scaling).  Give a multiple of {world}, or --batch-per-gpu")
        args.batch_per_gpu = args.global_batch // world
    if args.stub_step:
        return stub_main(args, json_fd)
    # Rehearsal (tests/test_rccl_gpu.py on a one-GPU box): all ranks share cuda:0 and the group is gloo -- everything the N > 1
    # step does except RCCL itself (rank offsets into the synthetic batch, the gathered-slab checks, max-over-ranks timing,
    # the JSON fields).  The line says so and is not a measurement.
    rehearsal = os.environ.get("BALF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if n_dev < (1 if rehearsal else max(world, local_rank + 1)):
        raise SystemExit(f"[bench] rank {rank}: {n_dev} GPU(s) visible, {world} ranks asked for: refusing to report "
                         f"n_gpus={world} from fewer devices")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: balf_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    collective_note = None
    have_group = False
    if world > 1:
        if "MASTER_PORT" not in os.environ:
            raise SystemExit("[bench] MASTER_PORT is not set: the ranks of one run must agree on it")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            _real_gather = dist.all_gather_into_tensor

            def _gather_through_host(out, inp, group=None):          # gloo moves host memory: stage the slabs there
                o = torch.empty(out.shape, dtype=out.dtype)
                _real_gather(o, inp.cpu(), group=group)
                out.copy_(o)
            dist.all_gather_into_tensor = _gather_through_host
            _real_all_gather = dist.all_gather

            def _all_gather_through_host(outs, inp, group=None):
                tmp = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
                _real_all_gather(tmp, inp.cpu(), group=group)
                for o, t_ in zip(outs, tmp):
                    o.copy_(t_)
            dist.all_gather = _all_gather_through_host
            collective_note = "REHEARSAL: gloo through host memory, all ranks on cuda:0 (not RCCL, not a measurement)"
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        have_group = True
    elif not args.no_single_rank_collective:
        # one GPU: run the path's collective anyway on a single-rank RCCL group (SURVEY.md 8e caveat), so that the step
        # timed here is the step the N > 1 runs time
        try:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            have_group = True
        except Exception as e:          # noqa: BLE001 -- report and go on without it
            collective_note = f"none (single-rank RCCL group failed: {type(e).__name__}: {e})"[:200]
    # the library must be the release build: a timing-ablation build (csrc/diag.h) computes wrong results and only loads
    # through an explicit BALF_HIP_LIB override, in which case the line says so instead of passing as a measurement
    from balf_amd import _lib as _balf_lib
    build_flags = _balf_lib.lib().balf_build_flags().decode()
    if not build_flags.startswith("release"):
        if not args.allow_diagnostic_build:
            raise SystemExit(f"[bench] {_balf_lib.LIB_PATH} is a DIAGNOSTIC build ({build_flags}): it computes wrong results; "
                             "refusing to measure it (--allow-diagnostic-build for a timing experiment: the line then says INVALID)")
        print(f"[bench] DIAGNOSTIC library build: {build_flags}", file=sys.stderr)
    h, w, k, b = args.height, args.width, args.topk, args.batch_per_gpu
    hp, wp, top, left = arch.padded_hw(h, w)
    state = synth.synthetic_state_dict(20240)
    model = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    model.load_state_dict(state)
    model.precision = args.precision
    model = model.eval().to(dev)

    def resident_input(gray_u8, hh, ww):
        """padded fp32 NCHW batch in HBM, as the reference's callers hand it to the model"""
        hp_, wp_, top_, left_ = arch.padded_hw(hh, ww)
        g = torch.from_numpy(gray_u8).to(dev).float().div_(255.0)
        x_ = torch.zeros((gray_u8.shape[0], 3, hp_, wp_), dtype=torch.float32, device=dev)
        x_[:, :, top_:top_ + hh, left_:left_ + ww] = g[:, None]
        return x_

    # which device and how much workspace every rank really uses (the 8-GPU run binds rank r to cuda:LOCAL_RANK and sizes its own
    # workspace; the rehearsal puts every rank on cuda:0 and says so)
    rank_info = {"rank": rank, "local_rank_env": int(os.environ.get("LOCAL_RANK", "0")), "cuda_device": dev.index,
                 "forward_workspace_bytes": int(_balf_lib.lib().balf_forward_workspace_bytes(b, hp, wp)),
                 "images": [rank * b, rank * b + b]}
    rank_devices = [rank_info]
    if have_group and world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, rank_info)

    # synthetic inputs, resident in HBM before the timed region: this rank's shard of the global batch
    gray = synthetic_batch(h, w, rank * b, b)
    x = resident_input(gray, h, w)

    def make_step(x_, hh, ww, kk, want_logits=False):
        _, _, top_, left_ = arch.padded_hw(hh, ww)

        def step():
            idx, score, count, prob = pipeline.detect_batch(model, x_, hh, ww, 15, 15, kk, precomputed_offsets=(top_, left_),
                                                            want_logits=want_logits)
            gi, gs, gc = pipeline.allgather_keypoints(idx, score, count, force=have_group)
            return gi, gs, gc, (idx, score, count, prob)
        return step

    def fence():
        torch.cuda.synchronize(dev)
        if have_group:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_run(step, steps, warmup):
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
        rank_dt[:] = [dt_, dt_]
        if have_group and world > 1:
            tmax = torch.tensor([dt_, -dt_], dtype=torch.float64, device=dev)     # max and (negated) min over ranks
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax[0].item())
            rank_dt[:] = [dt_, -float(tmax[1].item())]
        return dt_, prof_, o

    rank_dt = [0.0, 0.0]                    # slowest / fastest rank's time of the last timed_run

    def summarize(precision, dt_, prof_, steps, b=b, hp=hp, wp=wp):
        """images/s + the roofline of the dominant kernel against both roofs and the issue limit (b, hp, wp: the
        configuration the profile was taken on; default: the headline one)."""
        mb = _balf_lib.lib().balf_forward_micro_batch(b, hp, wp)      # images per launch (make_plan in det_common.h)
        name, (ms, n_launch) = max(prof_.items(), key=lambda kv: kv[1][0])
        avg_ms = ms / n_launch
        flops = kernel_flops_per_launch(name, mb, hp, wp)
        nbytes = kernel_bytes_per_launch(name, mb, hp, wp, precision_is_f16=(precision != "fp32"))
        peak_tf = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_FP16_MFMA_TFLOPS
        tf = flops / (avg_ms * 1e-3) / 1e12 if flops else 0.0
        gbs = nbytes / (avg_ms * 1e-3) / 1e9 if nbytes else 0.0
        f_m, f_h = tf / peak_tf, gbs / PEAK_HBM_GBS
        pmc = pmc_profile(precision, mb, hp, wp)
        slot = (pmc or {}).get(name)
        roof = {"kernel": name, "avg_launch_ms": avg_ms, "launches": n_launch, "images_per_launch": mb,
                "traffic": slot["hbm_bytes_per_launch"] if slot else None,
                "mfma": {"achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": f_m,
                         "algorithmic_flop_per_launch": flops,
                         "products_per_mac": 3 if precision == "fp16" else 1},
                "hbm": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_h,
                        "algorithmic_bytes_per_launch": nbytes}}
        if slot:
            # vector + matrix instructions of one launch priced at what a SIMD charges with >= 2 waves resident (2.6 cycles
            # per VALU instruction, 12.5 per 16x16x32 f16 MFMA beside vector work / 32 per 16x16x4 f32 MFMA;
            # tools/ubench/mfma_valu_mix.hip) over the SIMD cycles of the launch in the same profiled run
            roof["issue"] = {k_: slot[k_] for k_ in ("valu_insts", "mfma_insts", "issue_share", "valu_busy_share",
                                                     "mfma_busy_share", "valu_mfma_coexec_share", "wave_wait_share",
                                                     "wave_issue_stall_share", "waves") if k_ in slot}
        share = (slot or {}).get("issue_share", 0.0)
        if max(f_m, f_h) < 0.5 and share > max(f_m, f_h):
            roof.update({"bound": "valu_issue", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_h,
                         "note": "neither roof above 0.5: vector/matrix instruction issue is the limit (see issue.issue_share) -- at the "
                                 "clock the package power cap allows (power_state_under_load; the same instruction streams on zero "
                                 "operands hold 2.36 GHz of 2.4 at 1125 W and run 20 % faster: profiles/r5_power_zero.txt); "
                                 "achieved/peak/frac are the HBM figures"})
        elif f_m >= f_h:
            roof.update({"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": f_m})
            if precision == "fp32":
                # (round 5, profiles/r5_f32.txt) an fp32 MFMA holds the SIMD's vector pipe for its whole duration: matrix time and
                # vector time ADD (co-execution counter 0), so the fraction of the f32 MFMA roof a kernel with any vector work can
                # reach is mfma_busy / (mfma_busy + valu_busy)
                roof["note"] = ("fp32 MFMAs do not run beside vector instructions (tools/ubench/mfma_valu_mix_f32): the pipe both kinds "
                                "share is busy issue.mfma_busy_share + issue.valu_busy_share of the time")
        else:
            roof.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": f_h})
        fwd_ms = sum(v[0] for kname, v in prof_.items() if kname.startswith("stage")) / steps
        nms_ms = sum(v[0] for kname, v in prof_.items() if not kname.startswith("stage")) / steps
        out_ = {"images_per_s": world * b * steps / dt_, "ms_per_step": dt_ / steps * 1e3, "roofline": roof,
                "forward_device_ms_per_step": fwd_ms, "nms_topk_device_ms_per_step": nms_ms,
                "forward_tflops": b * hp * wp * arch.FLOP_PER_PADDED_PIXEL / (fwd_ms * 1e-3) / 1e12,
                "kernels_ms_per_step": {kname: v[0] / steps for kname, v in sorted(prof_.items())}}
        if pmc:
            tot = sum(v["hbm_bytes_per_launch"] for kname, v in pmc.items() if kname.startswith("stage"))
            px = mb * hp * wp
            out_["forward_hbm"] = {
                "measured_bytes_per_launch_group": tot, "images": mb, "bytes_per_padded_pixel": tot / px,
                "achieved_GBps": tot * (b / mb) / (fwd_ms * 1e-3) / 1e9,
                "frac_of_8TBps": tot * (b / mb) / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "reference_points_bytes_per_px": {"algorithmic_minimum": 20.0, "fused_schedule_plan_fp16": 1100.0},
                "source": os.path.relpath(PMC_PROFILE, ROOT)}
        return out_

    step = make_step(x, h, w, k)
    dt, prof, out = timed_run(step, args.steps, args.warmup)
    if model.effective_precision != args.precision:
        # the module runs a checkpoint that fails its split-f16 range check on the fp32 kernels (a warning for a caller,
        # but a number measured on the other path under this path's name for a benchmark)
        raise SystemExit(f"[bench] the model ran precision={model.effective_precision!r} instead of {args.precision!r} "
                         "(split-f16 range check failed): refusing to report")
    counts = out[2]
    kp_per_image = float(counts.float().mean().item())
    head = summarize(args.precision, dt, prof, args.steps)
    per_rank = {"min": b * args.steps / rank_dt[0], "max": b * args.steps / rank_dt[1]}
    # the same step with the logits written too (the reference's forward always returns them; the headline step skips the
    # 272 MB store nobody reads): a few steps, reported beside the headline
    dt_l, _, _ = timed_run(make_step(x, h, w, k, want_logits=True), max(args.steps // 2, 2), 1)
    logits_on = {"images_per_s": world * b * max(args.steps // 2, 2) / dt_l, "steps": max(args.steps // 2, 2)}
    logits_on["ratio_to_headline"] = logits_on["images_per_s"] / head["images_per_s"]

    # the collective by itself: device time of pack + all_gather_into_tensor + unpack on this rank's slabs, events on the
    # stream it runs on (RCCL is stream-ordered on torch's current stream)
    allgather_us = None
    if have_group:
        i_, s_, c_ = out[3][0], out[3][1], out[3][2]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            pipeline.allgather_keypoints(i_, s_, c_, force=True)
        fence()
        e0.record()
        for _ in range(20):
            pipeline.allgather_keypoints(i_, s_, c_, force=True)
        e1.record()
        fence()
        allgather_us = e0.elapsed_time(e1) * 1e3 / 20
    # every rank must end up with the SAME gathered slabs, its own shard at rows [rank * b, (rank + 1) * b)
    slabs_identical = None
    if have_group:
        gi_, gs_, gc_ = out[0], out[1], out[2]
        own = (torch.equal(gi_[rank * b:(rank + 1) * b], out[3][0]) and torch.equal(gc_[rank * b:(rank + 1) * b], out[3][2])
               and gi_.shape[0] == world * b)
        w8 = torch.arange(1, gi_.numel() + 1, device=dev, dtype=torch.int64).view(gi_.shape) % 1000003      # order-sensitive
        chk = torch.stack([(gi_.long() * w8).sum(), (gs_.view(torch.int32).long() * w8).sum(), gc_.long().sum(),
                           torch.tensor(int(own), device=dev)])
        allc = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        slabs_identical = bool(all(torch.equal(c_[:3], allc[0][:3]) for c_ in allc) and all(int(c_[3]) == 1 for c_ in allc))
        if not slabs_identical:
            raise SystemExit(f"[bench] rank {rank}: the gathered keypoint slabs differ between ranks (or a shard is misplaced)")
    # the CPU sample: images spread over the whole batch, so that both micro-batches of the forward are looked at
    n_cpu = min(max(args.cpu_images, 1), b)
    cpu_pick = sorted(set(int(round(i * (b - 1) / max(n_cpu - 1, 1))) for i in range(n_cpu)))
    pick_t = torch.tensor(cpu_pick, device=dev)
    gpu_sample = tuple(t.index_select(0, pick_t).clone() for t in out[3])

    other = None
    other_prec = "fp32" if args.precision == "fp16" else "fp16"
    if args.other_steps > 0:
        model.precision = other_prec
        dt2, prof2, _ = timed_run(step, args.other_steps, 1)
        other = summarize(other_prec, dt2, prof2, args.other_steps)
        other["precision"] = other_prec
        model.precision = args.precision

    # the other BASELINE configurations that fit one GPU (driver-timed too: a few steps each), each with the roofline of its
    # own dominant kernel and -- rank 0 -- a CPU baseline on two of its images (north_star: "throughput on synthetic
    # VGA/720p/1080p ... as absolute numbers and as fraction of the HBM/MFMA roofline, next to the reference CPU path")
    other_cfgs = None
    if world == 1 and args.other_configs and (h, w, b) == (1080, 1920, 32):
        other_cfgs = []
        del x
        cpu_done = {}
        for (name, bb, hh, ww, kk, prec, steps) in (("configs[1]: 32 x 640x480, top-1000", 32, 480, 640, 1000, "fp16", 5),
                                                    ("configs[2]: 64 x 1280x720, top-2000, fp32", 64, 720, 1280, 2000, "fp32", 2),
                                                    ("configs[2] on the split-f16 path", 64, 720, 1280, 2000, "fp16", 3),
                                                    ("configs[4]: 128 x 1920x1080 fp16, top-2000 (1 GPU)", 128, 1080, 1920, 2000, "fp16", 2)):
            model.precision = prec
            g8 = synthetic_batch(hh, ww, 0, min(bb, 8))
            xx = resident_input(np.tile(g8, ((bb + 7) // 8, 1, 1))[:bb], hh, ww)       # images 0..7, 0..7, ...
            st = make_step(xx, hh, ww, kk)
            d, pf, o = timed_run(st, steps, 1)
            hp_, wp_, top_, left_ = arch.padded_hw(hh, ww)
            sm = summarize(prec, d, pf, steps, b=bb, hp=hp_, wp=wp_)
            entry = {"workload": name, "precision": prec, "images_per_s": bb * steps / d,
                     "keypoints_per_s": bb * steps / d * float(o[2].float().mean().item()),
                     "ms_per_step": d / steps * 1e3, "steps": steps, "roofline": sm["roofline"],
                     "forward_tflops": sm["forward_tflops"], "kernels_ms_per_step": sm["kernels_ms_per_step"]}
            if args.cpu_images > 0 and (hh, ww) != (h, w):
                # the CPU oracle on the first two images of this configuration (once per size; 1080p has the headline's)
                if (hh, ww) not in cpu_done:
                    rep, cprobs, cdets = cpu_baseline(g8[:2], kk, state, args.cpu_threads)
                    cpu_done[(hh, ww)] = (rep, cprobs, cdets)
                rep, cprobs, cdets = cpu_done[(hh, ww)]
                entry["cpu_baseline"] = rep
                gpu_s = tuple(t[:2].clone() for t in o[3])
                entry["index_match"] = dict(index_match(gpu_s, cprobs, cdets, hh, ww, kk, top_, left_), precision=prec)
            other_cfgs.append(entry)
            del xx, st, o
        model.precision = args.precision
        torch.cuda.empty_cache()

    # batch-1 latency: the only way the reference ever calls the model (/root/reference/demo/demo_match.py:29,
    # balf/utils/train_utils.py:428) -- one uint8 image in, keypoints out, a synchronisation per call
    latency = power = None
    if world == 1 and args.other_configs and (h, w, b) == (1080, 1920, 32):
        power = power_state_under_load(step, dev)
        if power is not None:
            # the forward is bound by the package power limit: joule per image is the figure a kernel change has
            # to move -- package power under load / images per second of this rank
            power["joule_per_image"] = power["package_power_w"] / (head["images_per_s"] / world)
            power["note"] += "; joule_per_image = package power / this rank's images per second"
        latency = [single_image_latency(model, dev, hh, ww, kk) for (hh, ww, kk) in ((480, 640, 1000), (1080, 1920, 2000))]

    sustained = host_fed = natural = None
    rccl_ranks = dist.get_world_size() if have_group else 0
    if world == 1 and not rehearsal:
        if args.sustained_seconds > 0:
            sustained = sustained_run(step, dev, b, args.sustained_seconds)
            sustained["ratio_to_timed_steps"] = sustained["images_per_s"] / head["images_per_s"]
        if args.cpu_images > 0:
            natural = natural_match(model, dev)
        if args.host_fed_steps > 0:
            # a single-GPU leg with one host synchronisation per step: the single-rank RCCL group of the headline step is taken
            # down first -- while it is alive (torch's NCCL watchdog thread polling the runtime) this host-paced loop, and the
            # resident step timed beside it, lose ~4 % (measured: ratio 0.948 with the group, 0.986 without)
            if have_group:
                dist.barrier()
                dist.destroy_process_group()
                have_group = False
            host_fed = host_fed_run(model, dev, gray, k, args.host_fed_steps, step)
            host_fed["collective"] = "none (single-GPU leg; the headline's single-rank RCCL group is destroyed before it)"

    if rank == 0:
        ips = head["images_per_s"]
        nms_bytes = b * (4.0 * h * w + 8.0 * k + 4.0)
        nms_ms = head["nms_topk_device_ms_per_step"]
        dtype = ("f16 MFMA, split hi+lo operands (3 products), f32 accumulate/LN/GELU/softmax/NMS"
                 if args.precision == "fp16" else "f32")
        if collective_note is None:
            collective_note = ("all_gather_into_tensor of [B,2K+1] int32 keypoint slabs (RCCL)"
                               + ("" if world > 1 else ", single-rank group") if rccl_ranks else "none")
        res = {
            "metric": ("REHEARSAL (ranks share one GPU, gloo): not a measurement" if rehearsal else
                       "images/sec + keypoints/sec on 1080p gray (detector forward + NMS + top-K); NMS index match vs CPU ref"
                       if build_flags.startswith("release") else "INVALID: diagnostic library build (timing ablation, wrong results)"),
            "rehearsal": rehearsal,
            "value": ips, "unit": "images/s", "keypoints_per_s": ips * kp_per_image,
            "keypoints_per_image": kp_per_image,
            "n_gpus": world, "rccl_ranks": rccl_ranks,
            "per_rank_images_per_s": per_rank, "allgather_device_us": allgather_us,
            "gathered_slabs_identical": slabs_identical,
            "rank_devices": rank_devices,
            "rank_env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
            "launched_by": "bench.py" if os.environ.get("BALF_BENCH_LAUNCHED") else
                           ("external launcher" if "TORCHELASTIC_RUN_ID" in os.environ or world > 1 else "direct"),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{b} images/GPU x {world} GPU, {w}x{h} gray -> [B,3,{hp},{wp}] fp32, top-{k}, "
                                   f"border 15, nms 15 (BASELINE configs[3]/[4] shard)",
                       "global_batch": b * world, "parallelism": f"dp{world}", "precision": args.precision,
                       "collective": collective_note},
            "roofline": head["roofline"],
            "forward_device_ms_per_step": head["forward_device_ms_per_step"],
            "nms_topk_device_ms_per_step": nms_ms,
            "forward_tflops": head["forward_tflops"],
            "forward_hbm": head.get("forward_hbm"),
            "nms_topk_roofline": {"bound": "hbm", "achieved": nms_bytes / (nms_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": nms_bytes / (nms_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "kernels_ms_per_step": head["kernels_ms_per_step"],
            "other_precision": other,
            "other_configs": other_cfgs,
            "with_logits": logits_on,
            "sustained": sustained,
            "host_fed": host_fed,
            "batch1_latency": latency,
            "power_state_under_load": power,
            "library_build": build_flags,
        }
        if world == 1 and args.cpu_images > 0:
            res["cpu_baseline"], cpu_probs, cpu_dets = cpu_baseline(gray[cpu_pick], k, state, args.cpu_threads)
            res["cpu_baseline"]["sample"] = res["cpu_baseline"]["sample"].replace("(the first of the GPU batch)", f"(images {cpu_pick} of the GPU batch)")
            res["index_match"] = index_match(gpu_sample, cpu_probs, cpu_dets, h, w, k, top, left)
            res["index_match"]["precision"] = args.precision
            res["index_match"]["batch_images"] = cpu_pick
            res["index_match"]["natural"] = natural
            if other_cfgs is not None:
                # SURVEY 8(d)'s second CPU leg: the oracle with batch = min(B, 8) images per forward call, at VGA
                res["cpu_baseline"]["batched"] = cpu_baseline(synthetic_batch(480, 640, 0, 8), 1000, state, args.cpu_threads,
                                                              batch=8)[0]
        else:
            res["cpu_baseline"] = None
            res["index_match"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if have_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
