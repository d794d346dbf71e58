// DSnoT list-head kernel, n:m instantiations (templates: dsnot_lists.hpp).
#include "dsnot_lists.hpp"

namespace vlmc {

int dsnot_refine_lists_nm(const void *W, int dtype, int64_t out_f, int64_t in_f, int64_t ldw, const uint8_t *keep0, const float *sq,
                          const float *sum_row, const float *var_row, int use_wanda_init, int prune_m, int max_cycle, float thr,
                          float pow_var, int without_same_sign, uint32_t *events, int32_t *t_row, hipStream_t st) {
    return lists_dispatch_dtype<true>(W, dtype, out_f, in_f, ldw, keep0, sq, sum_row, var_row, use_wanda_init, prune_m, max_cycle,
                                      thr, pow_var, without_same_sign, events, t_row, st);
}

}  // namespace vlmc
