// Thread-local error text and ABI bookkeeping for libsparsify_hip.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

int sc_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* sc_last_error(void) { return g_err; }
extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }
extern "C" size_t sc_abi_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(sc_block_desc);
        case 1: return sizeof(sc_gemm_epilogue);
        default: return 0;
    }
}
extern "C" int sc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
