// ABI bookkeeping entry points of libnerfhip (include/nerfhip.h).
#include "common.h"

extern "C" int nerfhip_abi_version(void) { return NERFHIP_ABI_VERSION; }

extern "C" const char* nerfhip_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case NERFHIP_E_BADARG: return "nerfhip: bad argument (null pointer, non-positive size or unsupported shape)";
        case NERFHIP_E_UNSUPPORTED: return "nerfhip: unsupported dtype/architecture";
        case NERFHIP_E_ALIGN: return "nerfhip: pointer not aligned (16 bytes; 128 for the per-point buffers of the render kernels)";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "nerfhip: unknown error";
}
