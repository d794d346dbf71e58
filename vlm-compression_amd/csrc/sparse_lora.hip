// K14-K16: SparseLoRA (replaces the tensor algebra of
// /root/reference/lavis/peft/src/peft/tuners/lora.py:359-394).
//
//   forward  sparse=True : W_eff = (W + s*(B@A)) . M        (lora.py:362-368)
//            sparse=False: W_eff =  W . M + s*(B@A)          (lora.py:369-375)
//   merge()  sparse=True : W += (s*(B@A)) . M                (lora.py:385-387)
//            sparse=False: W[~M] = 0; W += s*(B@A)           (lora.py:388-391)
//   backward dB = ((dW_eff [. M]) * s) @ A^T,  dA = B^T @ ((dW_eff [. M]) * s)
//
// The [out,in] delta s*(B@A) is never written to HBM.  One workgroup owns a 32-row x 256-column
// tile: the rank-r contraction runs on the matrix cores with the f32-input MFMA (exact fp32,
// k-ordered fma chain => deterministic), the tile goes through LDS so that every global access
// is a coalesced 16-byte-per-lane row segment, and the mask / rounding chain of the reference
// is applied elementwise.  HBM-bound: effective weight = read W (2) + M (1) + write (2) B/weight;
// gradient = read G (2) + M (1) B/weight per pass.  The big GEMMs (x @ W_eff^T, dY @ W_eff,
// dY^T @ x) stay with the library (hipBLASLt through torch), as the tier rules prescribe.
//
// Rounding (weight dtype wd = fp16/bf16; identity for fp32), exactly the reference's op chain:
//   forward: d1 = wd(B@A); d2 = wd(d1 * s); sparse: wd(W + d2) * M; else wd(W*M + d2)
//   merge  : delta stays fp32; W = wd(float(W [*M]) + delta [*M])
// `ab_code` (0 none, 1 fp16, 2 bf16) is the autocast dtype: A, B and the product B@A are rounded to it,
// and so are the intermediate and final adapter gradients, as the reference's autograd does.
#include "common.hpp"

namespace vlmc {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<f32_t>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<f16_t>(float v) { return float(_Float16(v)); }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return float(__bf16(v)); }

template <typename T> __device__ __forceinline__ typename T::raw from_f32(float v);
template <> __device__ __forceinline__ float from_f32<f32_t>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t from_f32<f16_t>(float v) {
    _Float16 h = _Float16(v);
    uint16_t r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
template <> __device__ __forceinline__ uint16_t from_f32<bf16_t>(float v) {
    __bf16 h = __bf16(v);
    uint16_t r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}

// rounding to the autocast dtype of the reference's `B @ A` (0 = none / fp32, 1 = fp16, 2 = bf16)
__device__ __forceinline__ float round_code(float v, int code) {
    return code == 1 ? float(_Float16(v)) : (code == 2 ? float(__bf16(v)) : v);
}

enum { LORA_FWD_SPARSE = 0, LORA_FWD_MASKED = 1, LORA_MERGE_SPARSE = 2, LORA_MERGE_MASKED = 3 };

constexpr int kTM = 32;        // tile rows
constexpr int kTN = 256;       // tile columns (4 waves x 64)
constexpr int kLd = kTN + 1;   // LDS row stride in floats (odd => conflict-free column walks)

// ------------------------------------------------------------------------------------------
// effective weight / merge
// ------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void lora_weff_kernel(const typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f,
                                                        int64_t ldw, const float *__restrict__ A, const float *__restrict__ B,
                                                        int r, float scaling, const uint8_t *__restrict__ mask, int ab_code,
                                                        typename T::raw *__restrict__ Wout, int64_t ldo) {
    using raw = typename T::raw;
    __shared__ float dl[kTM * kLd];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = int64_t(blockIdx.y) * kTM, colb = int64_t(blockIdx.x) * kTN;

    // ---- delta tile on the matrix cores: D[o, i] = sum_k B[o, k] * A[k, i] -------------------------
    // v_mfma_f32_32x32x2_f32: A-operand lane l = Bm[o = l&31][k = l>>5], B-operand = Am[k = l>>5][i = l&31],
    // D register g of lane l = D[row (g&3) + 8*(g>>2) + 4*(l>>5)][col l&31].
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    const int64_t orow = row0 + l31;
    for (int k = 0; k < r; k += 2) {
        const int kk = k + h;
        float b = (orow < out_f && kk < r) ? B[orow * r + kk] : 0.f;
        b = round_code(b, ab_code);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t c = colb + wave * 64 + t * 32 + l31;
            float a = (c < in_f && kk < r) ? A[int64_t(kk) * in_f + c] : 0.f;
            a = round_code(a, ab_code);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[t], 0, 0, 0);
        }
    }
    constexpr bool kMerge = MODE == LORA_MERGE_SPARSE || MODE == LORA_MERGE_MASKED;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int o = (g & 3) + 8 * (g >> 2) + 4 * h;
            const int c = wave * 64 + t * 32 + l31;
            float d = acc[t][g];
            if constexpr (kMerge) {
                d = ieee_mul(d, scaling);                                   // fp32 delta (merge runs outside autocast)
            } else {
                d = round_to<T>(ieee_mul(round_to<T>(round_code(d, ab_code)), scaling));   // (B@A).to(wd) * scaling
            }
            dl[o * kLd + c] = d;
        }
    __syncthreads();

    // ---- elementwise combine, 16-byte chunks ----------------------------------------------------------
    const bool vec = (in_f % 8 == 0) && (ldw % 8 == 0) && (ldo % 8 == 0) && aligned16_dev(W) && aligned16_dev(Wout) &&
                     (reinterpret_cast<uintptr_t>(mask) % 8 == 0);
#pragma unroll
    for (int q4 = 0; q4 < (kTM * kTN / 8) / 256; ++q4) {
        const int q = tid + 256 * q4;
        const int o = q >> 5, cc = (q & 31) * 8;
        const int64_t row = row0 + o, col = colb + cc;
        if (row >= out_f || col >= in_f) continue;
        raw w[8];
        uint8_t m[8];
        if (vec) {
            Chunk8<T> c8 = load_chunk8<T>(W + row * ldw + col);
            __builtin_memcpy(w, c8.v, sizeof(w));
            const uint2 mm = *reinterpret_cast<const uint2 *>(mask + row * in_f + col);
            __builtin_memcpy(m, &mm, 8);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = col + j < in_f;
                w[j] = in ? W[row * ldw + col + j] : raw(0);
                m[j] = in ? mask[row * in_f + col + j] : uint8_t(0);
            }
        }
        Chunk8<T> res;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wf = to_f32<T>(w[j]);
            const float d = dl[o * kLd + cc + j];
            const bool keep = m[j] != 0;
            float v;
            if constexpr (MODE == LORA_FWD_SPARSE) v = keep ? round_to<T>(ieee_add(wf, d)) : 0.f;
            else if constexpr (MODE == LORA_FWD_MASKED) v = round_to<T>(ieee_add(keep ? wf : 0.f, d));
            else if constexpr (MODE == LORA_MERGE_SPARSE) v = round_to<T>(ieee_add(wf, keep ? d : 0.f));
            else v = round_to<T>(ieee_add(keep ? wf : 0.f, d));
            res.v[j] = from_f32<T>(v);
        }
        if (vec) {
            store_chunk8<T>(Wout + row * ldo + col, res);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (col + j < in_f) Wout[row * ldo + col + j] = res.v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------
// gradients of A and B.  Gm = (G [. M]) * s with the reference's rounding (wd(G*M) is exact,
// wd(. * s) rounds), staged per tile in LDS as fp32.
// v_mfma_f32_16x16x4_f32: A-operand lane l = X[i = l&15][k = l>>4], B-operand = Y[k = l>>4][j = l&15],
// D register g of lane l = D[row 4*(l>>4) + g][col l&15].
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void stage_gm_tile(const typename T::raw *__restrict__ G, int64_t out_f, int64_t in_f, int64_t ldg,
                                              const uint8_t *__restrict__ mask, int sparse, float scaling, int ab_code,
                                              int64_t row0, int64_t colb, float *gm, bool vec) {
    using raw = typename T::raw;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q4 = 0; q4 < (kTM * kTN / 8) / 256; ++q4) {
        const int q = tid + 256 * q4;
        const int o = q >> 5, cc = (q & 31) * 8;
        const int64_t row = row0 + o, col = colb + cc;
        float v[8];
        if (row < out_f && col < in_f) {
            raw g[8];
            uint8_t m[8];
            if (vec) {
                Chunk8<T> c8 = load_chunk8<T>(G + row * ldg + col);
                __builtin_memcpy(g, c8.v, sizeof(g));
                const uint2 mm = *reinterpret_cast<const uint2 *>(mask + row * in_f + col);
                __builtin_memcpy(m, &mm, 8);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = col + j < in_f;
                    g[j] = in ? G[row * ldg + col + j] : raw(0);
                    m[j] = in ? mask[row * in_f + col + j] : uint8_t(0);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = to_f32<T>(g[j]);
                if (sparse && !m[j]) x = 0.f;
                // autograd of `d1 * s` rounds to wd, the cast back to the matmul's dtype rounds to it
                v[j] = round_code(round_to<T>(ieee_mul(x, scaling)), ab_code);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) gm[o * kLd + cc + j] = v[j];
    }
}

// dB[o, k] = sum_i Gm[o, i] * A[k, i]: one workgroup per 32-row strip, loops over the columns.
// Wave w: rows (w&1)*16.., column half (w>>1)*128.. of each tile; NT16 = ceil(r/16) n-tiles.
template <typename T, int NT16>
__global__ __launch_bounds__(256) void lora_grad_b_kernel(const typename T::raw *__restrict__ G, int64_t out_f, int64_t in_f,
                                                          int64_t ldg, const float *__restrict__ A, int r, float scaling,
                                                          const uint8_t *__restrict__ mask, int sparse, int ab_code,
                                                          float *__restrict__ dB) {
    __shared__ float gm[kTM * kLd];
    __shared__ float red[2][kTM * 16 * NT16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int64_t row0 = int64_t(blockIdx.x) * kTM;
    const bool vec = (in_f % 8 == 0) && (ldg % 8 == 0) && aligned16_dev(G) && (reinterpret_cast<uintptr_t>(mask) % 8 == 0);
    const int rh = wave & 1, chalf = wave >> 1;
    f32x4 acc[NT16];
#pragma unroll
    for (int n = 0; n < NT16; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t colb = 0; colb < in_f; colb += kTN) {
        __syncthreads();
        stage_gm_tile<T>(G, out_f, in_f, ldg, mask, sparse, scaling, ab_code, row0, colb, gm, vec);
        __syncthreads();
        for (int i0 = chalf * 128; i0 < chalf * 128 + 128; i0 += 4) {
            const float x = gm[(rh * 16 + l15) * kLd + i0 + q];                 // Gm[o = l&15][i = i0 + (l>>4)]
            const int64_t ci = colb + i0 + q;
#pragma unroll
            for (int n = 0; n < NT16; ++n) {
                const int kk = n * 16 + l15;
                float a = (kk < r && ci < in_f) ? A[int64_t(kk) * in_f + ci] : 0.f;   // A^T[i][k]
                a = round_code(a, ab_code);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, a, acc[n], 0, 0, 0);
            }
        }
    }
    // combine the two column halves (fixed order => deterministic), write dB
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT16; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[chalf][(rh * 16 + 4 * q + g) * (16 * NT16) + n * 16 + l15] = acc[n][g];
    __syncthreads();
    for (int e = tid; e < kTM * 16 * NT16; e += 256) {
        const int o = e / (16 * NT16), kk = e % (16 * NT16);
        if (row0 + o < out_f && kk < r) {
            dB[(row0 + o) * r + kk] = round_code(ieee_add(red[0][e], red[1][e]), ab_code);
        }
    }
}

// dA[k, i] = sum_o B[o, k] * Gm[o, i]: one workgroup per 256-column strip, loops over the rows.
template <typename T, int NT16>
__global__ __launch_bounds__(256) void lora_grad_a_kernel(const typename T::raw *__restrict__ G, int64_t out_f, int64_t in_f,
                                                          int64_t ldg, const float *__restrict__ B, int r, float scaling,
                                                          const uint8_t *__restrict__ mask, int sparse, int ab_code,
                                                          int64_t rows_per_split, float *__restrict__ part) {
    // grid = (column strips, row splits); split s accumulates rows [s*rows_per_split, ...) into part[s][k][i]
    __shared__ float gm[kTM * kLd];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int64_t colb = int64_t(blockIdx.x) * kTN;
    const int64_t rbeg = int64_t(blockIdx.y) * rows_per_split;
    const int64_t rend = rbeg + rows_per_split < out_f ? rbeg + rows_per_split : out_f;
    float *dA = part + int64_t(blockIdx.y) * r * in_f;
    const bool vec = (in_f % 8 == 0) && (ldg % 8 == 0) && aligned16_dev(G) && (reinterpret_cast<uintptr_t>(mask) % 8 == 0);
    f32x4 acc[NT16][4];        // [k-tile][16-column tile of this wave's 64 columns]
#pragma unroll
    for (int n = 0; n < NT16; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[n][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t row0 = rbeg; row0 < rend; row0 += kTM) {
        __syncthreads();
        stage_gm_tile<T>(G, rend, in_f, ldg, mask, sparse, scaling, ab_code, row0, colb, gm, vec);
        __syncthreads();
        for (int o0 = 0; o0 < kTM; o0 += 4) {
            const int64_t orow = row0 + o0 + q;
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = gm[(o0 + q) * kLd + wave * 64 + c * 16 + l15];   // Gm[o = o0+(l>>4)][i]
#pragma unroll
            for (int n = 0; n < NT16; ++n) {
                const int kk = n * 16 + l15;
                float b = (kk < r && orow < rend) ? B[orow * r + kk] : 0.f;                      // B^T[k][o]
                b = round_code(b, ab_code);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[n][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, y[c], acc[n][c], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NT16; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kk = n * 16 + 4 * q + g;
                const int64_t ci = colb + wave * 64 + c * 16 + l15;
                if (kk < r && ci < in_f) dA[int64_t(kk) * in_f + ci] = acc[n][c][g];
            }
}

// dA = sum over the row splits, in split order (deterministic)
template <typename T>
__global__ void lora_grad_a_reduce_kernel(const float *__restrict__ part, int splits, int64_t n, int ab_code,
                                          float *__restrict__ dA) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v = ieee_add(v, part[int64_t(s) * n + i]);
    dA[i] = round_code(v, ab_code);
}

static int64_t grad_row_splits(int64_t out_f, int64_t in_f) {
    const int64_t strips = (in_f + kTN - 1) / kTN, tiles = (out_f + kTM - 1) / kTM;
    int64_t s = (512 + strips - 1) / strips;          // aim at >= 512 workgroups
    if (s > tiles) s = tiles;
    if (s < 1) s = 1;
    return s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <typename T>
static int weff_typed(const void *W, int64_t out_f, int64_t in_f, int64_t ldw, const float *A, const float *B, int r,
                      float scaling, const uint8_t *mask, int mode, int ab_code, void *Wout, int64_t ldo, hipStream_t st) {
    using raw = typename T::raw;
    const dim3 grid(unsigned((in_f + kTN - 1) / kTN), unsigned((out_f + kTM - 1) / kTM));
#define VLMC_WEFF(M)                                                                                              \
    hipLaunchKernelGGL((lora_weff_kernel<T, M>), grid, dim3(256), 0, st, static_cast<const raw *>(W), out_f, in_f, ldw, A, B, r, \
                       scaling, mask, ab_code, static_cast<raw *>(Wout), ldo)
    switch (mode) {
        case LORA_FWD_SPARSE: VLMC_WEFF(LORA_FWD_SPARSE); break;
        case LORA_FWD_MASKED: VLMC_WEFF(LORA_FWD_MASKED); break;
        case LORA_MERGE_SPARSE: VLMC_WEFF(LORA_MERGE_SPARSE); break;
        default: VLMC_WEFF(LORA_MERGE_MASKED); break;
    }
#undef VLMC_WEFF
    VLMC_HIP_CHECK_LAUNCH("vlmc_lora_effective_weight");
    return VLMC_OK;
}

template <typename T>
static int grad_typed(const void *G, int64_t out_f, int64_t in_f, int64_t ldg, const float *A, const float *B, int r,
                      float scaling, const uint8_t *mask, int sparse, int ab_code, float *dA, float *dB, float *ws,
                      hipStream_t st) {
    using raw = typename T::raw;
    const raw *g = static_cast<const raw *>(G);
    const unsigned gb = unsigned((out_f + kTM - 1) / kTM), ga = unsigned((in_f + kTN - 1) / kTN);
    const int64_t splits = grad_row_splits(out_f, in_f);
    const int64_t tiles = (out_f + kTM - 1) / kTM;
    const int64_t rows_per_split = (tiles + splits - 1) / splits * kTM;
#define VLMC_GRAD(NT)                                                                                                       \
    do {                                                                                                                    \
        if (dB) hipLaunchKernelGGL((lora_grad_b_kernel<T, NT>), dim3(gb), dim3(256), 0, st, g, out_f, in_f, ldg, A, r, scaling, \
                                   mask, sparse, ab_code, dB);                                        \
        if (dA) hipLaunchKernelGGL((lora_grad_a_kernel<T, NT>), dim3(ga, unsigned(splits)), dim3(256), 0, st, g, out_f, in_f, \
                                   ldg, B, r, scaling, mask, sparse, ab_code, rows_per_split, ws);               \
    } while (0)
    const int nt = (r + 15) / 16;
    switch (nt) {
        case 1: VLMC_GRAD(1); break;
        case 2: VLMC_GRAD(2); break;
        case 3: VLMC_GRAD(3); break;
        default: VLMC_GRAD(4); break;
    }
#undef VLMC_GRAD
    if (dA) {
        const int64_t n = int64_t(r) * in_f;
        hipLaunchKernelGGL((lora_grad_a_reduce_kernel<T>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, ws, int(splits), n,
                           ab_code, dA);
    }
    VLMC_HIP_CHECK_LAUNCH("vlmc_lora_grad");
    return VLMC_OK;
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_lora_effective_weight(const void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw,
                                          const float *A, const float *B, int r, float scaling, const uint8_t *mask, int mode,
                                          int ab_code, void *W_out, int64_t ldo, void *stream) {
    VLMC_REQUIRE(W && A && B && mask && W_out, "vlmc_lora_effective_weight: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && ldw >= in_features && ldo >= in_features && r > 0,
                 "vlmc_lora_effective_weight: bad shape out=%lld in=%lld r=%d", (long long)out_features,
                 (long long)in_features, r);
    VLMC_REQUIRE(mode >= 0 && mode <= 3, "vlmc_lora_effective_weight: unknown mode %d", mode);
    VLMC_REQUIRE(ab_code >= 0 && ab_code <= 2, "vlmc_lora_effective_weight: unknown autocast code %d", ab_code);
    VLMC_REQUIRE((out_features + kTM - 1) / kTM <= 65535, "vlmc_lora_effective_weight: too many rows");
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case VLMC_F32: return weff_typed<f32_t>(W, out_features, in_features, ldw, A, B, r, scaling, mask, mode, ab_code, W_out, ldo, st);
        case VLMC_F16: return weff_typed<f16_t>(W, out_features, in_features, ldw, A, B, r, scaling, mask, mode, ab_code, W_out, ldo, st);
        case VLMC_BF16: return weff_typed<bf16_t>(W, out_features, in_features, ldw, A, B, r, scaling, mask, mode, ab_code, W_out, ldo, st);
    }
    set_error("vlmc_lora_effective_weight: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}

extern "C" size_t vlmc_lora_grad_workspace(int64_t out_features, int64_t in_features, int r) {
    if (out_features <= 0 || in_features <= 0 || r <= 0) return 0;
    return round_up(size_t(grad_row_splits(out_features, in_features)) * size_t(r) * size_t(in_features) * 4, 256);
}

extern "C" int vlmc_lora_grad(const void *G, int dtype, int64_t out_features, int64_t in_features, int64_t ldg, const float *A,
                              const float *B, int r, float scaling, const uint8_t *mask, int sparse, int ab_code,
                              float *dA, float *dB, void *workspace, size_t workspace_bytes, void *stream) {
    VLMC_REQUIRE(G && A && B && mask, "vlmc_lora_grad: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && ldg >= in_features && r > 0 && r <= 64,
                 "vlmc_lora_grad: bad shape out=%lld in=%lld r=%d (r <= 64)", (long long)out_features, (long long)in_features, r);
    if (dA) {
        const size_t need = vlmc_lora_grad_workspace(out_features, in_features, r);
        VLMC_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) % 256) == 0, "vlmc_lora_grad: workspace missing or not 256-B aligned");
        if (workspace_bytes < need) {
            set_error("vlmc_lora_grad: workspace %zu B < required %zu B", workspace_bytes, need);
            return VLMC_EWORKSPACE;
        }
    }
    hipStream_t st = as_stream(stream);
    float *ws = static_cast<float *>(workspace);
    switch (dtype) {
        case VLMC_F32: return grad_typed<f32_t>(G, out_features, in_features, ldg, A, B, r, scaling, mask, sparse, ab_code, dA, dB, ws, st);
        case VLMC_F16: return grad_typed<f16_t>(G, out_features, in_features, ldg, A, B, r, scaling, mask, sparse, ab_code, dA, dB, ws, st);
        case VLMC_BF16: return grad_typed<bf16_t>(G, out_features, in_features, ldg, A, B, r, scaling, mask, sparse, ab_code, dA, dB, ws, st);
    }
    set_error("vlmc_lora_grad: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}
