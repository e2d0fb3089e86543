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

// Which stacks of BLSTM layers the recurrent kernels are built for (the reference takes any num_units per layer,
// models.py:95-99,107): any widths of 1 .. 256 units per direction, layer by layer (each padded to 256 in the packed layout,
// the padding's weights exactly zero).  More than 256 units: AVSI_ERR_UNSUPPORTED -- the host layer reports it through this
// entry point rather than by a rule of its own.
extern "C" int avsi_blstm_net_supported(const int* net_dim, int num_layers) {
    if (!net_dim || num_layers < 1) return AVSI_ERR_INVALID_ARG;
    for (int l = 0; l < num_layers; ++l) {
        if (net_dim[l] < 1) return AVSI_ERR_INVALID_ARG;
        if (net_dim[l] > 256) return AVSI_ERR_UNSUPPORTED;
    }
    return AVSI_OK;
}
