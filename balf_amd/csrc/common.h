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
