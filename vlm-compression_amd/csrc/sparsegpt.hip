// K10: SparseGPT blocked OBS sweep -- the sequential column loop of one 128-column block
// (replaces /root/reference/lavis/compression/pruners/sparsegpt_pruner.py:186-205; the Python loop
// there issues ~10 small kernels per column, ~20k launches per linear).
//
// Rows are independent inside a block, the columns are sequential:
//     for i in block:  [n:m: at i % m == 0 pick the n smallest w^2/d^2 of the next m columns]
//                      q = pruned ? 0 : w_i;  err = (w_i - q) / U[i,i];  w[i:] -= err * U[i, i:]
// Layout: one wave owns R rows at a time, LANES ARE COLUMNS (lane and lane+64 of the block), so the
// rank-1 update is one multiply + one subtract per lane, the pivot value travels by v_readlane, and
// the factor row U[i, :] is read from LDS once per step for all R rows.  The whole block factor
// (<= 128x128 fp32 = 64 KB) sits in LDS.  Every operation is an elementwise IEEE fp32 op in the
// reference's order (no fma: -ffp-contract=off), so given the same factor the sweep is bit-exact.
// The trailing update W[:, i2:] -= Err @ U[i1:i2, i2:] stays a library GEMM.
#include "common.hpp"

namespace vlmc {

constexpr int kSgBlock = 128;   // max columns per block
// rows per wave in flight (independent dependency chains) is the template parameter R: few rows per wave and
// more waves when the linear has few rows, so that every SIMD of the chip has a wave (2048 rows: 4 -> 2 rows per wave)

__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

template <int kSgRows>
__global__ __launch_bounds__(256) void sparsegpt_sweep_kernel(float *__restrict__ W, int64_t out_f, int count, int64_t ldw,
                                                              const float *__restrict__ U1, int64_t ldu,
                                                              const uint8_t *__restrict__ mask1, int64_t ldm, int prune_n,
                                                              int prune_m, float *__restrict__ Err1, int64_t lde,
                                                              uint8_t *__restrict__ mask_out, int64_t ldmo) {
    extern __shared__ __attribute__((aligned(16))) float sU[];   // [count][kSgBlock]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    for (int e = tid; e < count * kSgBlock; e += blockDim.x) {
        const int i = e / kSgBlock, j = e % kSgBlock;
        sU[e] = (j < count) ? U1[int64_t(i) * ldu + j] : 0.f;
    }
    __syncthreads();
    const bool c0 = lane < count, c1 = lane + 64 < count;
    const int64_t groups = (out_f + kSgRows - 1) / kSgRows;
    for (int64_t g = int64_t(blockIdx.x) * nwaves + wave; g < groups; g += int64_t(gridDim.x) * nwaves) {
        const int64_t r0 = g * kSgRows;
        float w0[kSgRows], w1[kSgRows], e0[kSgRows], e1[kSgRows];
        int m0[kSgRows], m1[kSgRows];
#pragma unroll
        for (int r = 0; r < kSgRows; ++r) {
            const int64_t row = r0 + r;
            const bool live = row < out_f;
            w0[r] = (live && c0) ? W[row * ldw + lane] : 0.f;
            w1[r] = (live && c1) ? W[row * ldw + lane + 64] : 0.f;
            m0[r] = (live && c0 && mask1) ? int(mask1[row * ldm + lane]) : 0;
            m1[r] = (live && c1 && mask1) ? int(mask1[row * ldm + lane + 64]) : 0;
            e0[r] = e1[r] = 0.f;
        }
        for (int i = 0; i < count; ++i) {
            const int li = i & 63;
            const bool hi_slot = i >= 64;
            if (prune_n != 0 && i % prune_m == 0) {
                // n smallest of w^2/d^2 over columns i..i+m-1 on the COMPENSATED weights (:190-192);
                // ties -> lowest column (stable), every lane computes the same ranks
#pragma unroll
                for (int r = 0; r < kSgRows; ++r) {
                    uint32_t t[8];       // order-preserving keys of the metric; NaN ranks last like torch.sort
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        if (a < prune_m && i + a < count) {
                            const int col = i + a;
                            const float wv = col >= 64 ? lane_bcast(w1[r], col & 63) : lane_bcast(w0[r], col & 63);
                            const float dv = sU[col * kSgBlock + col];
                            t[a] = score_key(ieee_div(ieee_mul(wv, wv), ieee_mul(dv, dv)));
                        } else {
                            t[a] = 0xFFFFFFFFu;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        if (a < prune_m && i + a < count) {
                            int rank = 0;
#pragma unroll
                            for (int b = 0; b < 8; ++b)
                                if (b < prune_m) rank += (t[b] < t[a] || (t[b] == t[a] && b < a)) ? 1 : 0;
                            const int col = i + a;
                            if (rank < prune_n && lane == (col & 63)) {
                                if (col >= 64) m1[r] = 1; else m0[r] = 1;
                            }
                        }
                    }
                }
            }
            const float h0 = sU[i * kSgBlock + lane], h1 = sU[i * kSgBlock + lane + 64];
            const float d = sU[i * kSgBlock + i];
            const bool upd0 = lane >= i, upd1 = lane + 64 >= i;       // columns >= i (:204)
#pragma unroll
            for (int r = 0; r < kSgRows; ++r) {
                const float wi = hi_slot ? lane_bcast(w1[r], li) : lane_bcast(w0[r], li);
                const int pr = hi_slot ? __builtin_amdgcn_readlane(m1[r], li) : __builtin_amdgcn_readlane(m0[r], li);
                const float q = pr ? 0.f : wi;
                const float err = ieee_div(wi - q, d);
                if (upd0) w0[r] = w0[r] - ieee_mul(err, h0);
                if (upd1) w1[r] = w1[r] - ieee_mul(err, h1);
                if (lane == li) {
                    if (hi_slot) { w1[r] = q; e1[r] = err; } else { w0[r] = q; e0[r] = err; }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kSgRows; ++r) {
            const int64_t row = r0 + r;
            if (row >= out_f) continue;
            if (c0) {
                W[row * ldw + lane] = w0[r];
                Err1[row * lde + lane] = e0[r];
                if (mask_out) mask_out[row * ldmo + lane] = uint8_t(m0[r]);
            }
            if (c1) {
                W[row * ldw + lane + 64] = w1[r];
                Err1[row * lde + lane + 64] = e1[r];
                if (mask_out) mask_out[row * ldmo + lane + 64] = uint8_t(m1[r]);
            }
        }
    }
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_sparsegpt_sweep(float *W, int64_t out_features, int64_t count, int64_t ldw, const float *U1, int64_t ldu,
                                    const uint8_t *mask1, int64_t ldm, int prune_n, int prune_m, float *Err1, int64_t lde,
                                    uint8_t *mask_out, int64_t ldmo, void *stream) {
    VLMC_REQUIRE(W && U1 && Err1, "vlmc_sparsegpt_sweep: null pointer");
    VLMC_REQUIRE(out_features > 0 && count > 0 && count <= kSgBlock && ldw >= count && ldu >= count && lde >= count,
                 "vlmc_sparsegpt_sweep: bad shape out=%lld count=%lld (max %d columns per block)", (long long)out_features,
                 (long long)count, kSgBlock);
    if (prune_n != 0) {
        VLMC_REQUIRE(prune_m > 0 && prune_m <= 8 && prune_n > 0 && prune_n <= prune_m,
                     "vlmc_sparsegpt_sweep: bad n:m = %d:%d (m <= 8)", prune_n, prune_m);
    } else {
        VLMC_REQUIRE(mask1 && ldm >= count, "vlmc_sparsegpt_sweep: unstructured pruning needs the block mask");
    }
    // rows per wave: as few as it takes to give every SIMD of the chip (1024) a wave, at most 4
    int rows = int(out_features / 1024);
    rows = rows < 1 ? 1 : (rows >= 4 ? 4 : (rows >= 2 ? 2 : 1));
    const int64_t groups = (out_features + rows - 1) / rows;
    int64_t grid = (groups + 3) / 4;
    if (grid > 512) grid = 512;
    const size_t lds = size_t(count) * kSgBlock * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        const int bytes = kSgBlock * kSgBlock * int(sizeof(float));
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sparsegpt_sweep_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(sparsegpt_sweep_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(sparsegpt_sweep_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            set_error("vlmc_sparsegpt_sweep: cannot reserve 64 KB of LDS");
            return VLMC_EHIP;
        }
        attr_set = true;
    }
#define VLMC_SWEEP(R)                                                                                                       \
    hipLaunchKernelGGL(sparsegpt_sweep_kernel<R>, dim3(unsigned(grid)), dim3(256), lds, as_stream(stream), W, out_features,  \
                       int(count), ldw, U1, ldu, mask1, ldm, prune_n, prune_m, Err1, lde, mask_out, ldmo)
    if (rows == 1) VLMC_SWEEP(1);
    else if (rows == 2) VLMC_SWEEP(2);
    else VLMC_SWEEP(4);
#undef VLMC_SWEEP
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparsegpt_sweep");
    return VLMC_OK;
}
