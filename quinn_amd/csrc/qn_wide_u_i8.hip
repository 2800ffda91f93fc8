// Second object of qn_wide_i8.hip (see QN_WIDE_PART there): the forward instances for relu / identity networks of 128 / 256 widths,
// k_i8_wide_fwd_u<KC, DP, LMIN, STASH>, and their launcher -- compiled beside the first so that the library builds in ~2 minutes.
#define QN_WIDE_PART 1
#include "qn_wide_i8.hip"
