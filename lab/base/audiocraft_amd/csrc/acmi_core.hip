// libacmi: version + thread-local error reporting (never throw across the C ABI).
#include "acmi_common.h"

#include <stdarg.h>

static thread_local char g_err[512] = "";

void acmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int acmi_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        acmi_set_error("%s: %s", what, hipGetErrorString(e));
        return ACMI_ELAUNCH;
    }
    return ACMI_OK;
}

extern "C" int acmi_version(void) { return ACMI_VERSION; }
extern "C" const char* acmi_last_error(void) { return g_err; }
