// Packed 2:4 weights ("an optional packed 2:4 format for inference"): a linear whose mask keeps exactly two
// of every four consecutive input columns (the n:m rule of wanda_pruner.py:326-329 at 2:4) stored as
//     values [out, in / 2]   the kept weights in column order
//     meta   [out, in / 8]   one byte per TWO groups: a 4-bit code per group, i0 | i1 << 2 with i0 < i1 the kept positions
// (the index encoding of the structured-sparse matrix instructions, V_SMFMAC: two 2-bit selectors per group of four).  9 / 16 of
// the dense bytes for 16-bit weights.  The kept positions come from the MASK, not from the values: a kept weight that happens
// to be zero stays a kept zero, so unpack(pack(W, mask)) == W . mask bit for bit and the mask itself is recovered.
// HBM-bound, one pass: a lane owns 8 columns (16 B of weights, 8 mask bytes) -> 8 B of values + 1 byte of meta.
#include "common.hpp"

namespace vlmc {

__global__ __launch_bounds__(256) void pack24_kernel(const uint16_t *__restrict__ W, int64_t ldw, const uint8_t *__restrict__ mask, int64_t ldm,
                                                     int64_t out_f, int64_t in_f, uint16_t *__restrict__ values, uint8_t *__restrict__ meta,
                                                     unsigned int *__restrict__ bad) {
    const int64_t chunks = in_f / 8, i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= out_f * chunks) return;
    const int64_t r = i / chunks, c = i - r * chunks;
    const uint16_t *w = W + r * ldw + c * 8;
    const uint8_t *m = mask + r * ldm + c * 8;
    uint16_t v[4] = {0, 0, 0, 0};
    unsigned code = 0, wrong = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int n = 0, idx[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m[4 * g + j]) {
                if (n < 2) idx[n] = j;
                ++n;
            }
        wrong |= n != 2;
        v[2 * g] = w[4 * g + idx[0]], v[2 * g + 1] = w[4 * g + idx[1]];
        code |= unsigned(idx[0] | idx[1] << 2) << (4 * g);
    }
    if (wrong) atomicAdd(bad, 1u);
    uint16_t *o = values + r * (in_f / 2) + c * 4;
    o[0] = v[0], o[1] = v[1], o[2] = v[2], o[3] = v[3];
    meta[r * chunks + c] = uint8_t(code);
}

__global__ __launch_bounds__(256) void unpack24_kernel(const uint16_t *__restrict__ values, const uint8_t *__restrict__ meta, int64_t out_f,
                                                       int64_t in_f, uint16_t *__restrict__ W, int64_t ldw, uint8_t *__restrict__ mask, int64_t ldm,
                                                       unsigned int *__restrict__ bad) {
    const int64_t chunks = in_f / 8, i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= out_f * chunks) return;
    const int64_t r = i / chunks, c = i - r * chunks;
    const uint16_t *v = values + r * (in_f / 2) + c * 4;
    const unsigned code = meta[r * chunks + c];
    uint16_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint8_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned wrong = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const unsigned i0 = (code >> (4 * g)) & 3u, i1 = (code >> (4 * g + 2)) & 3u;
        wrong |= i0 >= i1;                                                    // (never written by pack24_kernel)
        w[4 * g + i0] = v[2 * g], k[4 * g + i0] = 1;
        w[4 * g + i1] = v[2 * g + 1], k[4 * g + i1] = 1;
    }
    if (wrong) atomicAdd(bad, 1u);
    uint16_t *o = W + r * ldw + c * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = w[j];
    if (mask) {
        uint8_t *mo = mask + r * ldm + c * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) mo[j] = k[j];
    }
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_pack_24(const void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw, const uint8_t *mask, int64_t ldm,
                            void *values, uint8_t *meta, unsigned int *bad_groups, void *stream) {
    VLMC_REQUIRE(dtype == VLMC_F16 || dtype == VLMC_BF16, "vlmc_pack_24: dtype must be VLMC_F16 or VLMC_BF16");
    VLMC_REQUIRE(W && mask && values && meta && bad_groups, "vlmc_pack_24: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && in_features % 8 == 0, "vlmc_pack_24: in_features must be a positive multiple of 8");
    VLMC_REQUIRE(ldw >= in_features && ldm >= in_features && out_features * (in_features / 8) < (int64_t(1) << 39), "vlmc_pack_24: bad shape");
    const int64_t n = out_features * (in_features / 8);
    hipLaunchKernelGGL(pack24_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const uint16_t *>(W), ldw, mask,
                       ldm, out_features, in_features, static_cast<uint16_t *>(values), meta, bad_groups);
    VLMC_HIP_CHECK_LAUNCH("vlmc_pack_24");
    return VLMC_OK;
}

extern "C" int vlmc_unpack_24(const void *values, const uint8_t *meta, int dtype, int64_t out_features, int64_t in_features, void *W, int64_t ldw,
                              uint8_t *mask, int64_t ldm, unsigned int *bad_groups, void *stream) {
    VLMC_REQUIRE(dtype == VLMC_F16 || dtype == VLMC_BF16, "vlmc_unpack_24: dtype must be VLMC_F16 or VLMC_BF16");
    VLMC_REQUIRE(W && values && meta && bad_groups, "vlmc_unpack_24: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && in_features % 8 == 0, "vlmc_unpack_24: in_features must be a positive multiple of 8");
    VLMC_REQUIRE(ldw >= in_features && (!mask || ldm >= in_features) && out_features * (in_features / 8) < (int64_t(1) << 39), "vlmc_unpack_24: bad shape");
    const int64_t n = out_features * (in_features / 8);
    hipLaunchKernelGGL(unpack24_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const uint16_t *>(values), meta,
                       out_features, in_features, static_cast<uint16_t *>(W), ldw, mask, ldm, bad_groups);
    VLMC_HIP_CHECK_LAUNCH("vlmc_unpack_24");
    return VLMC_OK;
}
