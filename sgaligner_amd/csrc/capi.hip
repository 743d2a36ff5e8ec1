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

// Arithmetic mode of the MFMA kernels that have a split-precision variant (pointnet.hip): 0 = exact fp32 (default, every
// headline number), 1 = split-bf16 x3 (opt-in).  Initial value from SGA_MFMA_MODE=bf16x3 in the environment.
#include <stdlib.h>
#include <string.h>
static int g_mfma_mode = -1;
int sga_mfma_mode() {
    if (g_mfma_mode < 0) {
        const char* e = getenv("SGA_MFMA_MODE");
        g_mfma_mode = (e && strcmp(e, "bf16x3") == 0) ? 1 : 0;
    }
    return g_mfma_mode;
}
extern "C" int sga_set_mfma_mode(int mode) {
    const int old = sga_mfma_mode();
    if (mode != 0 && mode != 1) { sga_set_error("sga_set_mfma_mode: mode %d (0 = fp32, 1 = bf16x3)", mode); return -1; }
    g_mfma_mode = mode;
    return old;
}
extern "C" int sga_get_mfma_mode(void) { return sga_mfma_mode(); }
