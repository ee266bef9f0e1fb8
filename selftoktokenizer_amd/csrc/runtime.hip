// Error plumbing + version for libselftok_hip.so (C ABI, no exceptions cross the boundary).
#include "common.h"
#include "selftok_hip.h"   // the C ABI declared there must match the definitions below
#include <stdio.h>
#include <string.h>

namespace selftok {
static thread_local char g_err[512] = "";

void set_last_error(const char* msg)
{
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return SELFTOK_EHIP;
    }
    return SELFTOK_OK;
}
}  // namespace selftok

extern "C" {
const char* selftok_last_error(void) { return selftok::g_err; }
int selftok_version(void) { return 100; }  // 0.1.0
}
