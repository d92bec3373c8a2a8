// C-ABI plumbing shared by every entry point: error strings, launch checks, version.
#include "leclip_common.h"
#include <stdarg.h>

namespace {
thread_local char g_err[512] = "";
}

void leclip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int leclip_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        leclip_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LECLIP_E_LAUNCH;
    }
    return LECLIP_OK;
}

extern "C" int leclip_abi_version(void) { return LECLIP_ABI_VERSION; }

extern "C" const char* leclip_last_error(void) { return g_err; }

extern "C" const char* leclip_strerror(int code) {
    switch (code) {
        case LECLIP_OK: return "ok";
        case LECLIP_E_INVALID: return "invalid argument";
        case LECLIP_E_UNSUPPORTED: return "unsupported shape or dtype";
        case LECLIP_E_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}
