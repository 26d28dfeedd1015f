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

// 32-bit fill as a KERNEL of the library's own (nms_topk.hip).  Not hipMemsetAsync: under hipGraph capture (torch.cuda.graph) on
// ROCm 7.0 the runtime executed the memset at capture time instead of recording it, so a replayed pipeline never cleared its
// counters again (second replay: duplicated survivors, an out-of-bounds store) -- a kernel node is recorded like every other launch.
int balf_fill_u32(void *dst_dev, unsigned value, size_t n_words, hipStream_t stream);
