// Optional per-kernel timing (hipEvent pairs on the launch stream), used by bench.py for the roofline
// figures.  Off by default; when off a launch costs one predictable branch.  Measurement aid only:
// global state, not re-entrant.
#pragma once
#include <hip/hip_runtime.h>

namespace balf_prof {
enum Slot {
    // stage s in 0..3: 4*s + {0 grid branch, 1 block branch, 2 squeeze-excite, 3 pool (stage 4: head)}
    kNmsTile = 16,
    kTopkSelect = 17,
    kHnConv2 = 18,          // HardNet descriptor: conv1+conv2 (fused), conv3..conv6, final 8x8 GEMM
    kHnConv3 = 19,
    kHnConv4 = 20,
    kHnConv5 = 21,
    kHnConv6 = 22,
    kHnFc = 23,
    kPatchPyr = 24,         // demo path: pyramid level, patch sampling, descriptor matching
    kPatchSample = 25,
    kMatchNN = 26,
    kMatchMutual = 27,
    kGreedyKeep = 28,       // greedy NMS of the demo path
    kGreedyKill = 29,
    kNumSlots = 30,
};
extern bool g_on;
void before(int slot, hipStream_t st);
void after(hipStream_t st);
// Between chain_begin() and chain_end() the caller issues its launches back to back on one stream with nothing else in
// between (the detector forward): consecutive kernels then share one time stamp -- the end of kernel k is the start of
// kernel k + 1 -- instead of two events per launch.
void chain_begin();
void chain_end();
struct Chain {
    Chain() { if (g_on) chain_begin(); }
    ~Chain() { if (g_on) chain_end(); }
};
}  // namespace balf_prof

#define BALF_PROF(slot, stream, launch_stmt)                       \
    do {                                                           \
        if (balf_prof::g_on) balf_prof::before((slot), (stream));  \
        launch_stmt;                                               \
        if (balf_prof::g_on) balf_prof::after((stream));           \
    } while (0)
