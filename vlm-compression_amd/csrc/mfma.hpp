// Matrix-core helpers shared by the GEMM engine (gemm_nt.hip) and the batched attention products (attn_matmul.hip):
// v_mfma_f32_16x16x32_{bf16,f16} on 16-byte operand fragments, and the one rounding of an fp32 accumulator to the
// 16-bit output dtype.
#pragma once
#include "common.hpp"

namespace vlmc {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ f32x4_t mfma16(const u32x4_t &a, const u32x4_t &b, const f32x4_t &c);
template <> __device__ __forceinline__ f32x4_t mfma16<bf16_t>(const u32x4_t &a, const u32x4_t &b, const f32x4_t &c) {
    bf16x8_t x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16<f16_t>(const u32x4_t &a, const u32x4_t &b, const f32x4_t &c) {
    f16x8_t x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ uint16_t from_f32(float v);
template <> __device__ __forceinline__ uint16_t from_f32<bf16_t>(float v) {
    const __bf16 h = static_cast<__bf16>(v);          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    uint16_t r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
template <> __device__ __forceinline__ uint16_t from_f32<f16_t>(float v) {
    const _Float16 h = static_cast<_Float16>(v);
    uint16_t r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}

}  // namespace vlmc
