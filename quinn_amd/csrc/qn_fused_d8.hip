// Second object of qn_fused.hip (see QN_FUSED_PART there): the float64-MFMA fused kernels' instances for networks with 5..16
// inputs -- k_fused_bwd_f64<H, NH, 8 | 16, UNB> (gradient, any activation) and k_fused_fwd_f64<H, G, relu / identity, 8 | 16> -- compiled
// beside the first object so that the library's build time stays where it was.
#define QN_FUSED_PART 1
#include "qn_fused.hip"
