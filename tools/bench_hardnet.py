#!/usr/bin/env python3
"""Time the HardNet descriptor kernels (per-kernel device ms through balf_profile_*) and report accuracy.
Usage: python tools/bench_hardnet.py [n_patches] [reps]"""
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
    sd64 = {k: v.double() for k, v in synth.synthetic_hardnet_state_dict(515).items()}
    ref = oracle.hardnet_forward(sd64, base[:256].cpu().double()).numpy()
    print("max-abs descriptor error vs fp64 oracle:", float(np.abs(d[:256].cpu().numpy() - ref).max()))


if __name__ == "__main__":
    main()
