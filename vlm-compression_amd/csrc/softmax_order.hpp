// The ONE order in which this library forms a softmax row (vlmc_softmax_rows in row_reduce.hip and the fused attention in
// attn_fused.hip give a row the same bits):
//   mx    = max_j x_j
//   e_j   = exp2( (x_j - mx) * log2(e) )                 one IEEE subtraction, one IEEE multiplication, v_exp_f32 (1 ulp)
//   s_c   = ((e_c + e_{c+16}) + e_{c+32}) + ..          the 16 CLASSES c = j mod 16, each in ascending j
//   total = butterfly over the classes: pairs c ^ 1, then c ^ 2, c ^ 4, c ^ 8 (additions commute: both partners of a pair get the
//           same bits, so every class ends with the same total)
//   y_j   = e_j * (1 / total)                            (one IEEE division per row, one multiplication per entry; rounded once
//                                                         more if the output is 16-bit)
// (round 5, first version: expf and a division per entry -- ~45 VALU instructions per score, which is what the fused kernel then
// spent its time on; the row's probabilities are rounded to 16 bits right away in every caller, 1 ulp of fp32 is not seen.)
// Padding invariance: entries whose e is exactly 0 (masked at the dtype's minimum) BEHIND a row's live entries append "+ 0" to
// every class chain -- the row's bits do not depend on how far it was padded.  Why 16 classes: an accumulator of
// v_mfma_f32_16x16x32 holding S^T = K Q^T has, per lane, 4 consecutive keys of a 16-key tile (class 4 g + i for lane group g,
// register i), so the fused kernel forms the class sums with one add per accumulator register and finishes with two
// in-register levels and two cross-lane ones.
#pragma once
#include "common.hpp"

namespace vlmc {

__device__ __forceinline__ float softmax_exp(float x, float mx) {
    float t = ieee_mul(ieee_add(x, -mx), 1.44269504088896340736f);
    asm volatile("" : "+v"(t));                                        // (the product is a value of its own: no fused multiply-add)
    return __builtin_amdgcn_exp2f(t);                                   // exp2(-inf) = 0: masked and padded entries
}
__device__ __forceinline__ float softmax_inv(float total) { return ieee_div(1.0f, total); }
__device__ __forceinline__ float softmax_prob(float e, float inv) {
    float y = ieee_mul(e, inv);
    asm volatile("" : "+v"(y));                                        // (rounded to fp32 before any rounding to 16 bits)
    return y;
}

// 16 consecutive lanes hold the classes 0..15 of one row
__device__ __forceinline__ float softmax_class_max(float m) {
#pragma unroll
    for (int off = 1; off <= 8; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    return m;
}
__device__ __forceinline__ float softmax_class_tree(float s) {
#pragma unroll
    for (int off = 1; off <= 8; off <<= 1) s = ieee_add(s, __shfl_xor(s, off, kWave));
    return s;
}

}  // namespace vlmc
