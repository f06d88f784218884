#include "common.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void mmvae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* mmvae_error_string() { return g_err; }

int mmvae_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mmvae_set_error("%s: %s", what, hipGetErrorString(e));
        return MMVAE_EHIP;
    }
    return MMVAE_OK;
}
