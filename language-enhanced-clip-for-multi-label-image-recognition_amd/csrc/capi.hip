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

thread_local int g_leclip_walk_order = -1;
int leclip_walk_order() { return g_leclip_walk_order; }
extern "C" int leclip_set_walk_order(int order) {
    const int prev = g_leclip_walk_order;
    g_leclip_walk_order = order < 0 ? -1 : (order ? 1 : 0);
    return prev;
}

thread_local int g_leclip_gemm_family = -1;
int leclip_gemm_family() { return g_leclip_gemm_family; }
extern "C" int leclip_set_gemm_family(int family) {
    const int prev = g_leclip_gemm_family;
    g_leclip_gemm_family = (family == 128 || family == 256 || family == 384) ? family : -1;
    return prev;
}

extern "C" const char* leclip_strerror(int code) {
    switch (code) {
        case LECLIP_OK: return "ok";
        case LECLIP_E_INVALID: return "invalid argument";
        case LECLIP_E_UNSUPPORTED: return "unsupported shape or dtype";
        case LECLIP_E_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

#ifdef LECLIP_DIAG
unsigned long long* g_leclip_wglog = nullptr;
unsigned g_leclip_wglog_cap = 0, g_leclip_wglog_seq = 0;
// diagnostic library only: device buffer of 2 + 4 * cap u64 (zeroed by the caller) that the GEMM / attention workgroups append to
extern "C" void leclip_diag_set_wglog(void* device_buf, unsigned cap) { g_leclip_wglog = (unsigned long long*)device_buf; g_leclip_wglog_cap = cap; g_leclip_wglog_seq = 0; }
#endif
