#!/usr/bin/env python3
"""Time the HardNet descriptor kernels (per-kernel device ms through balf_profile_*) and report accuracy.
Usage: python tools/bench_hardnet.py [n_patches] [reps] [fp16-split|fp16]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balf_amd._lib import lib                                            # noqa: E402
from balf_amd.third_party.hardnet.hardnet_pytorch import HardNet          # noqa: E402
from balf_amd.utils import synth                                          # noqa: E402

MAC_PER_PATCH = (9 * 32 * 1024 + 9 * 32 * 32 * 1024 + 9 * 32 * 64 * 256 + 9 * 64 * 64 * 256 + 9 * 64 * 128 * 64
                 + 9 * 128 * 128 * 64 + 64 * 128 * 128)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    m = HardNet()
    m.precision = sys.argv[3] if len(sys.argv) > 3 else "fp16-split"
    m.load_state_dict(synth.synthetic_hardnet_state_dict(515))
    m = m.eval().to("cuda:0")
    base = synth.synthetic_patches(2048, 3).to("cuda:0")
    x = base.repeat((n + 2047) // 2048, 1, 1, 1)[:n].contiguous()
    with torch.inference_mode():
        d = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            d = m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    l = lib()
    ns = l.balf_profile_num_slots()
    ms = (C.c_float * ns)()
    cnt = (C.c_int * ns)()
    l.balf_profile_begin()
    with torch.inference_mode():
        m(x)
    l.balf_profile_end(ms, cnt)
    print(f"{n} patches: {dt * 1e3:.3f} ms  ({n / dt / 1e6:.3f} M patches/s, "
          f"{2 * MAC_PER_PATCH * n / dt / 1e12:.1f} algorithmic TFLOP/s, x3 products = {6 * MAC_PER_PATCH * n / dt / 1e12:.1f} f16 TFLOP/s)")
    for i in range(ns):
        if cnt[i] and l.balf_profile_slot_name(i).decode().startswith("hardnet"):
            print(f"  {l.balf_profile_slot_name(i).decode():18s} {ms[i]:8.3f} ms  ({cnt[i]} launches)")
    from oracle import oracle
    sd = synth.synthetic_hardnet_state_dict(515)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = oracle.hardnet_forward(sd64, base[:256].cpu().double()).numpy()
    err = float(np.abs(d[:256].cpu().numpy() - ref).max())
    print("max-abs descriptor error vs fp64 oracle:", err)
    # the reference's CPU path beside it (oracle port: same torch CPU ops as third_party/hardnet/hardnet_pytorch.py)
    import json
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    xc = base[:1000].cpu()                       # one demo_match chunk (demo_match.py:73-76)
    oracle.hardnet_forward(sd, xc[:64])
    t0 = time.perf_counter()
    oracle.hardnet_forward(sd, xc)
    cpu_dt = time.perf_counter() - t0
    dom = max((i for i in range(ns) if cnt[i] and l.balf_profile_slot_name(i).decode().startswith("hardnet")), key=lambda i: ms[i])
    mac = {"hardnet_conv1_2": 9 * 32 * 1024 + 9 * 32 * 32 * 1024, "hardnet_conv3": 9 * 32 * 64 * 256,
           "hardnet_conv4": 9 * 64 * 64 * 256, "hardnet_conv5": 9 * 64 * 128 * 64, "hardnet_conv6": 9 * 128 * 128 * 64,
           "hardnet_fc": 64 * 128 * 128}[l.balf_profile_slot_name(dom).decode()]
    ach = 2 * mac * n / (ms[dom] * 1e-3) / 1e12
    print(json.dumps({
        "metric": "HardNet descriptors/s (demo path, 32x32 patches -> 128-d)", "value": n / dt, "unit": "patches/s",
        "n_patches": n, "ms": dt * 1e3,
        "dtype": ("f16 MFMA, split hi+lo operands (3 products), f32 accumulate" if m.precision == "fp16-split"
                  else "f16 MFMA, plain f16 operands, f32 accumulate"),
        "max_abs_err_vs_fp64": err,
        "roofline": {"kernel": l.balf_profile_slot_name(dom).decode(), "bound": "mfma", "achieved": ach,
                     "peak": 2500.0 / (3 if m.precision == "fp16-split" else 1), "unit": "TFLOP/s",
                     "frac": ach / (2500.0 / (3 if m.precision == "fp16-split" else 1)),
                     "note": "algorithmic FLOP (one product per MAC) against the f16 dense peak / products per MAC"},
        "cpu_baseline": {"value": 1000 / cpu_dt, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
                         "sample": f"1000 patches, oracle.hardnet_forward (torch CPU fp32), {cpu_dt:.2f} s"}}))


if __name__ == "__main__":
    main()
