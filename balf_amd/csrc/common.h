// Shared helpers for the gfx950 kernels of libbalf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/balf_hip.h"

#define BALF_LAUNCH_CHECK()                      \
    do {                                         \
        if (hipGetLastError() != hipSuccess)     \
            return BALF_ERR_LAUNCH;              \
    } while (0)

static inline int balf_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t balf_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// 32-bit fill as a KERNEL of the library's own (nms_topk.hip).  Not hipMemsetAsync: a memset node captured from it into a hipGraph
// (torch.cuda.graph) on ROCm 7.0 clears correctly on the FIRST replay and writes a garbage value from the second replay on
// (tools/memset_capture_probe.py: 0, then 2046820352) -- the NMS counters of a replayed pipeline then sent stores out of bounds.
// A kernel node carries its arguments by value like every other launch.
int balf_fill_u32(void *dst_dev, unsigned value, size_t n_words, hipStream_t stream);
