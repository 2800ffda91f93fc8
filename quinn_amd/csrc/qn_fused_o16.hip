// Third object of qn_fused.hip (see QN_FUSED_PART there): the float64-MFMA fused gradient kernel's instances for networks with 5..16
// outputs, k_fused_bwd_f64<H, NH, 4 | 16, UNB, OWIDE>.
#define QN_FUSED_PART 2
#include "qn_fused.hip"
