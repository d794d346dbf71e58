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

// One column step for the rows of a wave.  HI: the pivot column lives in the upper register (columns 64..127);
// then the lower one is final and is left alone.
template <int kSgRows, bool HI>
__device__ __forceinline__ void sweep_step(int i, int lane, float h0, float h1, float d, float (&w0)[kSgRows], float (&w1)[kSgRows],
                                           float (&e0)[kSgRows], float (&e1)[kSgRows], const int (&m0)[kSgRows],
                                           const int (&m1)[kSgRows]) {
    const int li = i & 63;
#pragma unroll
    for (int r = 0; r < kSgRows; ++r) {
        const float wi = lane_bcast(HI ? w1[r] : w0[r], li);
        const int pr = __builtin_amdgcn_readlane(HI ? m1[r] : m0[r], li);
        const float q = pr ? 0.f : wi;
        const float err = ieee_div(wi - q, d);
        if (!HI) {
            w0[r] = lane > li ? w0[r] - ieee_mul(err, h0) : (lane == li ? q : w0[r]);      // columns >= i (:204); the pivot becomes q
            w1[r] = w1[r] - ieee_mul(err, h1);
            if (lane == li) e0[r] = err;
        } else {
            w1[r] = lane > li ? w1[r] - ieee_mul(err, h1) : (lane == li ? q : w1[r]);
            if (lane == li) e1[r] = err;
        }
    }
}

// n of every m columns on the COMPENSATED weights (:190-192): ranks of w^2/d^2 over columns i..i+m-1, ties -> lowest
// column (stable), every lane computes the same ranks
template <int kSgRows, int M>     // M: compile-time group size (4, 8) or 0 = prune_m at run time (<= 8)
__device__ __forceinline__ void nm_decide(int i, int count, int lane, int prune_n, int prune_m_rt, const float *sU,
                                          const float (&w0)[kSgRows], const float (&w1)[kSgRows], int (&m0)[kSgRows], int (&m1)[kSgRows]) {
    const int prune_m = M ? M : prune_m_rt;
    constexpr int kMax = M ? M : 8;
#pragma unroll
    for (int r = 0; r < kSgRows; ++r) {
        uint32_t t[kMax];       // order-preserving keys of the metric; NaN ranks last like torch.sort
#pragma unroll
        for (int a = 0; a < kMax; ++a) {
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
        for (int a = 0; a < kMax; ++a) {
            if (a < prune_m && i + a < count) {
                int rank = 0;
#pragma unroll
                for (int b = 0; b < kMax; ++b)
                    if (b < prune_m) rank += (t[b] < t[a] || (t[b] == t[a] && b < a)) ? 1 : 0;
                const int col = i + a;
                if (rank < prune_n && lane == (col & 63)) {
                    if (col >= 64) m1[r] = 1; else m0[r] = 1;
                }
            }
        }
    }
}

template <int kSgRows, int NM>     // NM: 0 = unstructured (block mask given), 4 / 8 = n:m with that m, 1 = n:m, m at run time
__global__ __launch_bounds__(256) void sparsegpt_sweep_kernel(float *__restrict__ W, int64_t out_f, int count, int64_t ldw,
                                                              const float *__restrict__ U1, int64_t ldu,
                                                              const uint8_t *__restrict__ mask1, int64_t ldm, int prune_n,
                                                              int prune_m, float *__restrict__ Err1, int64_t lde,
                                                              uint8_t *__restrict__ mask_out, int64_t ldmo) {
    extern __shared__ __attribute__((aligned(16))) float sU[];   // [count][kSgBlock]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    // stage the factor block: all of a thread's loads are issued before the first LDS store
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (count == kSgBlock && (ldu & 3) == 0 && aligned16_dev(U1)) {
        constexpr int kPer = kSgBlock * (kSgBlock / 4) / 256;      // 16 float4 per thread
        f32x4 v[kPer];
#pragma unroll
        for (int b = 0; b < kPer; ++b) {
            const int e = tid + b * 256;
            v[b] = *reinterpret_cast<const f32x4 *>(U1 + int64_t(e >> 5) * ldu + (e & 31) * 4);
        }
#pragma unroll
        for (int b = 0; b < kPer; ++b) {
            const int e = tid + b * 256;
            *reinterpret_cast<f32x4 *>(sU + (e >> 5) * kSgBlock + (e & 31) * 4) = v[b];
        }
    } else {
        for (int e0 = tid; e0 < count * kSgBlock; e0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int e = e0 + b * 256, i = e / kSgBlock, j = e % kSgBlock;
                v[b] = (e < count * kSgBlock && j < count) ? U1[int64_t(i) * ldu + j] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int e = e0 + b * 256;
                if (e < count * kSgBlock) sU[e] = v[b];
            }
        }
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
        float h0n = sU[lane], h1n = sU[lane + 64], dn = sU[0];      // factor row of the NEXT step, read one step ahead
        const int half = count < 64 ? count : 64;
        for (int i = 0; i < half; ++i) {
            if (NM && i % (NM > 1 ? NM : prune_m) == 0) nm_decide<kSgRows, (NM > 1 ? NM : 0)>(i, count, lane, prune_n, prune_m, sU, w0, w1, m0, m1);
            const float h0 = h0n, h1 = h1n, d = dn;
            if (i + 1 < count) {
                h0n = sU[(i + 1) * kSgBlock + lane];
                h1n = sU[(i + 1) * kSgBlock + lane + 64];
                dn = sU[(i + 1) * kSgBlock + i + 1];
            }
            sweep_step<kSgRows, false>(i, lane, h0, h1, d, w0, w1, e0, e1, m0, m1);
        }
        for (int i = 64; i < count; ++i) {
            if (NM && i % (NM > 1 ? NM : prune_m) == 0) nm_decide<kSgRows, (NM > 1 ? NM : 0)>(i, count, lane, prune_n, prune_m, sU, w0, w1, m0, m1);
            const float h1 = h1n, d = dn;
            if (i + 1 < count) {
                h1n = sU[(i + 1) * kSgBlock + lane + 64];
                dn = sU[(i + 1) * kSgBlock + i + 1];
            }
            sweep_step<kSgRows, true>(i, lane, 0.f, h1, d, w0, w1, e0, e1, m0, m1);
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
    static PerDeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        const int bytes = kSgBlock * kSgBlock * int(sizeof(float));
        bool ok = true;
#define VLMC_SWEEP_ATTR(R, NMV)                                                                                          \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(sparsegpt_sweep_kernel<R, NMV>),                         \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
        VLMC_SWEEP_ATTR(1, 0); VLMC_SWEEP_ATTR(2, 0); VLMC_SWEEP_ATTR(4, 0);
        VLMC_SWEEP_ATTR(1, 1); VLMC_SWEEP_ATTR(2, 1); VLMC_SWEEP_ATTR(4, 1);
        VLMC_SWEEP_ATTR(1, 4); VLMC_SWEEP_ATTR(2, 4); VLMC_SWEEP_ATTR(4, 4);
        VLMC_SWEEP_ATTR(1, 8); VLMC_SWEEP_ATTR(2, 8); VLMC_SWEEP_ATTR(4, 8);
#undef VLMC_SWEEP_ATTR
        if (!ok) {
            set_error("vlmc_sparsegpt_sweep: cannot reserve 64 KB of LDS");
            return VLMC_EHIP;
        }
        once.mark(dev);
    }
#define VLMC_SWEEP(R, NMV)                                                                                                       \
    hipLaunchKernelGGL((sparsegpt_sweep_kernel<R, NMV>), dim3(unsigned(grid)), dim3(256), lds, as_stream(stream), W, out_features, \
                       int(count), ldw, U1, ldu, mask1, ldm, prune_n, prune_m, Err1, lde, mask_out, ldmo)
#define VLMC_SWEEP_ROWS(NMV)          \
    do {                             \
        if (rows == 1) VLMC_SWEEP(1, NMV);      \
        else if (rows == 2) VLMC_SWEEP(2, NMV); \
        else VLMC_SWEEP(4, NMV);     \
    } while (0)
    if (prune_n == 0) VLMC_SWEEP_ROWS(0);
    else if (prune_m == 4) VLMC_SWEEP_ROWS(4);
    else if (prune_m == 8) VLMC_SWEEP_ROWS(8);
    else VLMC_SWEEP_ROWS(1);
#undef VLMC_SWEEP_ROWS
#undef VLMC_SWEEP
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparsegpt_sweep");
    return VLMC_OK;
}
