// C-ABI plumbing shared by every entry point: version, thread-local error text.
#include <stdarg.h>

#include "sga_common.h"

static thread_local char g_err[512] = "";

void sga_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sga_last_error(void) { return g_err; }
extern "C" int sga_version(void) { return 100; }  // 0.1.0
extern "C" int sga_device_cus(void) { return sga_num_cus(); }
