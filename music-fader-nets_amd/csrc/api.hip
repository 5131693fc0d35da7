// api.hip - version / error strings of the C ABI (include/fadernets.h).
#include "common.h"

extern "C" {

int fn_version(void) { return 6; }

const char* fn_strerror(int code) {
    switch (code) {
        case FN_OK: return "ok";
        case FN_E_NULL: return "required pointer is NULL";
        case FN_E_SHAPE: return "unsupported or inconsistent sizes";
        case FN_E_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        case FN_E_WORKSPACE: return "workspace too small";
        case FN_E_COUNT: return "too many scans in one call";
        case FN_E_UNSUPPORTED: return "single-launch path not eligible for this device / shape (nothing enqueued)";
        case FN_E_COMM: return "librccl could not be loaded (or lacks a symbol): fn_comm_* unavailable";
        default: break;
    }
    if (code >= FN_COMM_ERROR_BASE) return fn_comm_strerror(code);
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

}  // extern "C"
