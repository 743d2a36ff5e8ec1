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

// Arithmetic mode of the MFMA kernels that have a reduced-precision variant: 0 = exact fp32 (default, every headline number),
// 1 = split-bf16 x3 (opt-in: pointnet.hip, sweepb.hip), 2 = fp16 inputs / fp32 accumulate for WIDE loss tables and the ranking
// (opt-in: BASELINE.json configs[4]; wide16.hip, simrank.hip -- chosen by the Python layer, 100-d tables stay exact fp32).
// Initial value from SGA_MFMA_MODE=bf16x3|f16 in the environment.
#include <stdlib.h>
#include <string.h>
static int g_mfma_mode = -1;
int sga_mfma_mode() {
    if (g_mfma_mode < 0) {
        const char* e = getenv("SGA_MFMA_MODE");
        g_mfma_mode = (e && strcmp(e, "bf16x3") == 0) ? 1 : (e && strcmp(e, "f16") == 0) ? 2 : (e && strcmp(e, "f16x2") == 0) ? 3 : (e && strcmp(e, "f16x2p") == 0) ? 4 : 0;
    }
    return g_mfma_mode;
}
extern "C" int sga_set_mfma_mode(int mode) {
    const int old = sga_mfma_mode();
    if (mode < 0 || mode > 4) { sga_set_error("sga_set_mfma_mode: mode %d (0 = fp32, 1 = bf16x3, 2 = f16 for wide tables, 3 = f16x2, 4 = f16x2p)", mode); return -1; }
    g_mfma_mode = mode;
    return old;
}
extern "C" int sga_get_mfma_mode(void) { return sga_mfma_mode(); }
