// placeholder replaced below by the MFMA path
#include "qn_common.h"
bool qn_fused_supported(const qn_desc*, int, int, int, int) { return false; }
size_t qn_fused_workspace(const qn_desc*, int, int, int, int) { return 0; }
int qn_fused_run(const qn_desc*, int, const void*, const void*, const void*, const int32_t*, int, int, int, double*,
                 void*, void*, void*, size_t, hipStream_t) {
    qn_set_error("fused path not built");
    return QN_EUNSUPPORTED;
}
