// ABI bookkeeping for libosi_hip.so.
#include "osi_common.h"

extern "C" {
int osi_abi_version(void) { return 1; }
const char* osi_build_arch(void) { return "gfx950"; }
const char* osi_strerror(int code) {
    switch (code) {
        case OSI_OK: return "ok";
        case OSI_ERR_ARG: return "invalid argument (shape, pointer or alignment precondition)";
        case OSI_ERR_LAUNCH: return "HIP launch failed";
        case OSI_ERR_STATE: return "executor called out of order";
        default: return "unknown error";
    }
}
}
