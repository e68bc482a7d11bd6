// r2l_error.hip — last-error plumbing for the C ABI (include/r2l_hip.h: r2l_last_error).
#include "r2l_common.h"
#include <stdio.h>

static thread_local char g_err[512] = "";
thread_local const float* g_r2l_c2w_dev = nullptr;
thread_local r2l_config g_r2l_cfg = {};  // the config of the *_cfg call in progress on this thread (r2l_common.h)

void r2l_set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s -> %s (%d)", what, hipGetErrorString(e), (int)e);
}

void r2l_set_error_msg(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

extern "C" const char* r2l_last_error(void) { return g_err; }
