// Library-level entry points: ABI version and the thread-local error text.
#include "common.h"
#include <cstring>

namespace mhe {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mhe

// 2: mhe_conv_desc grew `tile` and `res_half` (read by every convolution entry) and mhe_conv_wgrad_nhwc takes the descriptor
// 3: statistic accumulators are fixed-point mhe_stat_t words (order-independent integer atomics); pixel counts are double
extern "C" int mhe_abi_version(void) { return 4; }
extern "C" const char *mhe_last_error(void) { return mhe::g_err; }
