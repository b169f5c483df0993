// ABI-level helpers of libset_hip.so (version, error strings).
#include "set_common.h"

extern "C" {

int set_abi_version(void) { return 1; }

const char* set_error_string(int code) {
    switch (code) {
        case SET_OK: return "ok";
        case SET_ERR_ARG: return "invalid argument (null pointer, non-positive size or misaligned tensor)";
        case SET_ERR_UNSUPPORTED: return "dimension not supported by the gfx950 kernels";
        case SET_ERR_HIP: return "HIP runtime error (see set_last_hip_error_string)";
        case SET_ERR_WORKSPACE: return "workspace too small";
        case SET_ERR_FAULT: return "an earlier persistent launch (caption encoder / small-batch decode loop) timed out waiting for its workgroups (its outputs were poisoned with NaN); per-step kernels from now on";
        default: return "unknown error";
    }
}

int set_last_hip_error(void) { return set::g_last_hip_error; }
const char* set_last_hip_error_string(void) { return hipGetErrorString((hipError_t)set::g_last_hip_error); }
const char* set_target_arch(void) { return "gfx950"; }

}  // extern "C"
