// Diagnostic build switches of libbalf_hip.so, ALL in this one header.
//
// Every switch below either produces WRONG RESULTS (timing ablations: work or synchronisation is left out) or adds
// instrumentation to the kernels.  They exist for the experiments DESIGN.md quotes (tools/build_variant.sh builds a
// separately named library for each) and must never reach the shipped library:
//   * a translation unit with any of them set does not compile unless -DBALF_ALLOW_DIAGNOSTIC_BUILD=1 is given too;
//   * balf_build_flags() (include/balf_hip.h) returns the values compiled into the library, "release" first when all are
//     off -- tests/test_build_invariants.py and bench.py refuse a library that says anything else.
#pragma once

#ifndef BALF_ABLATE_GELU
#define BALF_ABLATE_GELU 0       // no GELU (det_common.h, stage1_f16.h, stage_cs_f16.h)
#endif
#ifndef BALF_ABLATE_BARRIER
#define BALF_ABLATE_BARRIER 0    // no workgroup barriers in the channel-split kernels (detector_f16.hip: lds_barrier)
#endif
#ifndef BALF_ABLATE_LOADLAT
#define BALF_ABLATE_LOADLAT 0    // activation rows come from a small L2-resident window (detector_f16.hip: load_frag_px)
#endif
#ifndef BALF_ABLATE_LUTCOPY
#define BALF_ABLATE_LUTCOPY 0    // the GELU table is not copied into LDS (stage_cs_f16.h)
#endif
#ifndef BALF_ABLATE_WSTREAM
#define BALF_ABLATE_WSTREAM 0    // every weight tile is the same L1-resident KiB (stage_cs_f16.h: CsBlob::frag)
#endif
#ifndef BALF_ABLATE_SPLIT
#define BALF_ABLATE_SPLIT 0      // operand split without the residual half (split16.h)
#endif
#ifndef BALF_DROP_WLO
#define BALF_DROP_WLO 0          // accuracy experiment: no (weight lo) x (activation hi) product (split16.h)
#endif
#ifndef BALF_STAMPS
#define BALF_STAMPS 0            // per-phase s_memtime stamps in the detector kernels (detector_f16.hip)
#endif
#ifndef BALF_HN_STAMPS
#define BALF_HN_STAMPS 0         // the same in the HardNet kernels (hardnet.hip)
#endif
#ifndef BALF_ABLATE_QSTREAM
#define BALF_ABLATE_QSTREAM 0    // stage-2 block kernel: RSHMAG.dense2's weights are read from LDS (another Linear's tiles) instead of streamed from L2
#endif
#ifndef BALF_DEBUG_STOP
#define BALF_DEBUG_STOP 0        // the f16 forward stops after the stage named by the environment variable BALF_DEBUG_STOP_STAGE (tests/experiments/s2_debug.py)
#endif
#ifndef BALF_F32_DBG
#define BALF_F32_DBG 0           // exact-fp32 stage-1 grid kernel (stage1_f32.h) stores intermediate tensor k into U; with BALF_DEBUG_STOP_STAGE set the forward stops there (tests/experiments/f32_s1_debug.py)
#endif
#ifndef BALF_ABLATE_UWINDOW
#define BALF_ABLATE_UWINDOW 0    // stages 1-2: u' (grid kernel's store, block kernel's load) wraps inside a window of this many MiB (a power of two), i.e. stays in the Infinity Cache: the bound of "u' through the cache" (round 6)
#endif
#ifndef BALF_S1_STRICT
#define BALF_S1_STRICT 0         // every hand-placed vmcnt wait of the persistent kernels drains the queue (debugging aid: correct, slow)
#endif

#define BALF_DIAGNOSTIC_BUILD                                                                                      \
    (BALF_ABLATE_GELU || BALF_ABLATE_BARRIER || BALF_ABLATE_LOADLAT || BALF_ABLATE_LUTCOPY || BALF_ABLATE_WSTREAM || \
     BALF_ABLATE_SPLIT || BALF_DROP_WLO || BALF_STAMPS || BALF_HN_STAMPS || BALF_S1_STRICT || BALF_DEBUG_STOP || BALF_ABLATE_QSTREAM || BALF_F32_DBG || \
     BALF_ABLATE_UWINDOW)
#if BALF_DIAGNOSTIC_BUILD && !defined(BALF_ALLOW_DIAGNOSTIC_BUILD)
#error "a diagnostic switch (csrc/diag.h) is set: pass -DBALF_ALLOW_DIAGNOSTIC_BUILD=1 as well (tools/build_variant.sh does) -- such a library must not ship"
#endif

#define BALF_DIAG_STR2(x) #x
#define BALF_DIAG_STR(x) BALF_DIAG_STR2(x)
#define BALF_DIAG_ITEM(name) " " #name "=" BALF_DIAG_STR(name)
#define BALF_DIAG_FLAGS_STRING                                                                                     \
    BALF_DIAG_ITEM(BALF_ABLATE_GELU) BALF_DIAG_ITEM(BALF_ABLATE_BARRIER) BALF_DIAG_ITEM(BALF_ABLATE_LOADLAT)        \
    BALF_DIAG_ITEM(BALF_ABLATE_LUTCOPY) BALF_DIAG_ITEM(BALF_ABLATE_WSTREAM) BALF_DIAG_ITEM(BALF_ABLATE_SPLIT)       \
    BALF_DIAG_ITEM(BALF_DROP_WLO) BALF_DIAG_ITEM(BALF_STAMPS) BALF_DIAG_ITEM(BALF_HN_STAMPS) BALF_DIAG_ITEM(BALF_S1_STRICT) \
    BALF_DIAG_ITEM(BALF_DEBUG_STOP) BALF_DIAG_ITEM(BALF_ABLATE_QSTREAM) BALF_DIAG_ITEM(BALF_F32_DBG) BALF_DIAG_ITEM(BALF_ABLATE_UWINDOW)
