// Library-level entry points of libbalf_hip.so (version, error strings, device check).
#include "common.h"
#include "diag.h"

#include <string.h>

extern "C" int balf_abi_version(void) { return BALF_ABI_VERSION; }

extern "C" const char *balf_error_string(int code) {
    switch (code) {
        case BALF_OK: return "ok";
        case BALF_ERR_ARG: return "bad argument (null pointer, non-positive size or unsupported parameter)";
        case BALF_ERR_SHAPE: return "bad shape (H/W not a multiple of 64, crop outside the map, or K > H*W)";
        case BALF_ERR_WORKSPACE: return "workspace too small";
        case BALF_ERR_ARCH: return "current device is not gfx950 (MI355X)";
        case BALF_ERR_LAUNCH: return "HIP launch failed";
        default: return "unknown error";
    }
}

extern "C" int balf_device_check(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return BALF_ERR_ARCH;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return BALF_ERR_ARCH;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? BALF_OK : BALF_ERR_ARCH;
}

// What kind of build this is (csrc/diag.h): "release ..." when every diagnostic switch is off.
extern "C" const char *balf_build_flags(void) {
#if BALF_DIAGNOSTIC_BUILD
    return "DIAGNOSTIC" BALF_DIAG_FLAGS_STRING;
#else
    return "release" BALF_DIAG_FLAGS_STRING;
#endif
}
