// ABI bookkeeping entry points of libavsi_hip.so.
#include "avsi_common.h"

extern "C" int avsi_abi_version(void) { return AVSI_ABI_VERSION; }

extern "C" const char* avsi_status_string(int status) {
    switch (status) {
        case AVSI_OK: return "ok";
        case AVSI_ERR_INVALID_ARG: return "invalid argument";
        case AVSI_ERR_UNSUPPORTED: return "unsupported shape for the gfx950 kernels";
        case AVSI_ERR_LAUNCH: return "kernel launch failed";
        case AVSI_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}
