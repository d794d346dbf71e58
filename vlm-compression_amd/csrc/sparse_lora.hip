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
constexpr int kLd = kTN + 4;   // LDS row stride in floats: rows stay 16-byte aligned (ds_read_b128), +4 skews the banks

// ------------------------------------------------------------------------------------------
// effective weight / merge
// ------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void lora_weff_kernel(const typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f,
                                                        int64_t ldw, const float *__restrict__ A, const float *__restrict__ B,
                                                        int r, float scaling, const uint8_t *__restrict__ mask, int ab_code,
                                                        typename T::raw *__restrict__ Wout, int64_t ldo) {
    using raw = typename T::raw;
    __shared__ __attribute__((aligned(16))) float dl[kTM * kLd];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = int64_t(blockIdx.y) * kTM, colb = int64_t(blockIdx.x) * kTN;

    // ---- the tile's W / mask chunks are requested first: their latency hides behind the MFMA phase -------------
    const bool vec = (in_f % 8 == 0) && (ldw % 8 == 0) && (ldo % 8 == 0) && aligned16_dev(W) && aligned16_dev(Wout) &&
                     (reinterpret_cast<uintptr_t>(mask) % 8 == 0);
    constexpr int NQ = (kTM * kTN / 8) / 256;
    Chunk8<T> wpre[NQ];
    uint2 mpre[NQ];
    if (vec) {
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
            const int q = tid + 256 * q4;
            const int64_t row = row0 + (q >> 5), col = colb + (q & 31) * 8;
            const bool in = row < out_f && col < in_f;
            wpre[q4] = load_chunk8<T>(W + (in ? row * ldw + col : 0));
            mpre[q4] = *reinterpret_cast<const uint2 *>(mask + (in ? row * in_f + col : 0));
        }
    }

    // ---- delta tile on the matrix cores: D[o, i] = sum_k B[o, k] * A[k, i] -------------------------
    // v_mfma_f32_32x32x2_f32: A-operand lane l = Bm[o = l&31][k = l>>5], B-operand = Am[k = l>>5][i = l&31],
    // D register g of lane l = D[row (g&3) + 8*(g>>2) + 4*(l>>5)][col l&31].
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    const int64_t orow = row0 + l31;
    // operands of 16 rank steps are fetched before the first MFMA (a load -> MFMA chain per step is pure latency)
    constexpr int KC = 16;
    for (int k0 = 0; k0 < r; k0 += KC) {
        float bv[KC / 2], av[KC / 2][2];
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; ++s2) {
            const int kk = k0 + 2 * s2 + h;
            bv[s2] = (orow < out_f && kk < r) ? B[orow * r + kk] : 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t c = colb + wave * 64 + t * 32 + l31;
                av[s2][t] = (c < in_f && kk < r) ? A[int64_t(kk) * in_f + c] : 0.f;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; ++s2) {
            const float b = round_code(bv[s2], ab_code);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, round_code(av[s2][t], ab_code), acc[t], 0, 0, 0);
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
#pragma unroll
    for (int q4 = 0; q4 < NQ; ++q4) {
        const int q = tid + 256 * q4;
        const int o = q >> 5, cc = (q & 31) * 8;
        const int64_t row = row0 + o, col = colb + cc;
        if (row >= out_f || col >= in_f) continue;
        raw w[8];
        uint8_t m[8];
        if (vec) {
            __builtin_memcpy(w, wpre[q4].v, sizeof(w));
            __builtin_memcpy(m, &mpre[q4], 8);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = col + j < in_f;
                w[j] = in ? W[row * ldw + col + j] : raw(0);
                m[j] = in ? mask[row * in_f + col + j] : uint8_t(0);
            }
        }
        Chunk8<T> res;
        float dv[8];
        {
            const float4 d0 = *reinterpret_cast<const float4 *>(&dl[o * kLd + cc]);
            const float4 d1 = *reinterpret_cast<const float4 *>(&dl[o * kLd + cc + 4]);
            dv[0] = d0.x; dv[1] = d0.y; dv[2] = d0.z; dv[3] = d0.w; dv[4] = d1.x; dv[5] = d1.y; dv[6] = d1.z; dv[7] = d1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wf = to_f32<T>(w[j]);
            const float d = dv[j];
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
// A 32 x 256 tile of G and of the mask in registers (4 chunks of 8 columns per thread), so that the NEXT tile's
// global loads are in flight while the current tile is consumed from LDS.
template <typename T> struct GmRegs {
    Chunk8<T> g[(kTM * kTN / 8) / 256];
    uint2 m[(kTM * kTN / 8) / 256];
};
template <typename T>
__device__ __forceinline__ void load_gm_tile(const typename T::raw *__restrict__ G, int64_t row_end, int64_t in_f, int64_t ldg,
                                             const uint8_t *__restrict__ mask, int64_t row0, int64_t colb, bool vec,
                                             GmRegs<T> &rg) {
    using raw = typename T::raw;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q4 = 0; q4 < (kTM * kTN / 8) / 256; ++q4) {
        const int q = tid + 256 * q4;
        const int64_t row = row0 + (q >> 5), col = colb + (q & 31) * 8;
        const bool in = row < row_end && col < in_f;
        if (vec) {
            rg.g[q4] = load_chunk8<T>(G + (in ? row * ldg + col : 0));
            rg.m[q4] = *reinterpret_cast<const uint2 *>(mask + (in ? row * in_f + col : 0));
        } else {
            uint8_t m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool inj = in && col + j < in_f;
                rg.g[q4].v[j] = inj ? G[row * ldg + col + j] : raw(0);
                m[j] = inj ? mask[row * in_f + col + j] : uint8_t(0);
            }
            __builtin_memcpy(&rg.m[q4], m, 8);
        }
    }
}
template <typename T>
__device__ __forceinline__ void store_gm_tile(const GmRegs<T> &rg, int64_t row_end, int64_t in_f, int sparse, float scaling,
                                              int ab_code, int64_t row0, int64_t colb, float *gm) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int q4 = 0; q4 < (kTM * kTN / 8) / 256; ++q4) {
        const int q = tid + 256 * q4;
        const int o = q >> 5, cc = (q & 31) * 8;
        const int64_t row = row0 + o, col = colb + cc;
        uint8_t m[8];
        __builtin_memcpy(m, &rg.m[q4], 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = to_f32<T>(rg.g[q4].v[j]);
            if (sparse && !m[j]) x = 0.f;
            // autograd of `d1 * s` rounds to wd, the cast back to the matmul's dtype rounds to it
            v[j] = (row < row_end && col + j < in_f) ? round_code(round_to<T>(ieee_mul(x, scaling)), ab_code) : 0.f;
        }
        *reinterpret_cast<float4 *>(&gm[o * kLd + cc]) = float4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<float4 *>(&gm[o * kLd + cc + 4]) = float4{v[4], v[5], v[6], v[7]};
    }
}

// One pass over G for both gradients.  Workgroup (strip j, split s) walks the 32-row tiles of rows
// [s*rows_per_split, ...) of the 256-column strip j; per tile (staged once in LDS as Gm):
//   dB tile [32, r] = Gm[32, 256] @ A_strip^T   -> part_b[j][row][k]   (summed over the strips by the reduce kernel)
//   dA strip [r, 256] += B_tile^T[r, 32] @ Gm   -> registers, written once as part_a[s][k][col]
// The A strip panel is staged in LDS once, the B tile per row tile: no global load sits inside an MFMA chain.
// Partial sums are combined in a fixed order (no float atomics) => deterministic.
template <typename T, int NT16>
__global__ __launch_bounds__(256) void lora_grad_fused_kernel(const typename T::raw *__restrict__ G, int64_t out_f, int64_t in_f,
                                                              int64_t ldg, const float *__restrict__ A,
                                                              const float *__restrict__ B, int r, float scaling,
                                                              const uint8_t *__restrict__ mask, int sparse, int ab_code,
                                                              int64_t rows_per_split, float *__restrict__ part_a,
                                                              float *__restrict__ part_b) {
    constexpr int RK = NT16 * 16;                    // padded rank
    __shared__ __attribute__((aligned(16))) float gm[kTM * kLd];
    __shared__ float As[RK * kLd];
    __shared__ float Bs[kTM * (RK + 1)];
    __shared__ float red[2][kTM * RK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int64_t colb = int64_t(blockIdx.x) * kTN;
    const int64_t rbeg = int64_t(blockIdx.y) * rows_per_split;
    const int64_t rend = rbeg + rows_per_split < out_f ? rbeg + rows_per_split : out_f;
    const bool vec = (in_f % 8 == 0) && (ldg % 8 == 0) && aligned16_dev(G) && (reinterpret_cast<uintptr_t>(mask) % 8 == 0);
    const int rh = wave & 1, chalf = wave >> 1;
    for (int e = tid; e < RK * kTN; e += 256) {
        const int kk = e / kTN, c = e % kTN;
        const int64_t ci = colb + c;
        As[kk * kLd + c] = (kk < r && ci < in_f) ? round_code(A[int64_t(kk) * in_f + ci], ab_code) : 0.f;
    }
    f32x4 acc_a[NT16][4];        // [k-tile][16-column tile of this wave's 64 columns]
#pragma unroll
    for (int n = 0; n < NT16; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc_a[n][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float *pb = part_b ? part_b + int64_t(blockIdx.x) * out_f * r : nullptr;
    GmRegs<T> rg;
    float breg[(kTM * RK + 255) / 256];
    auto load_b_tile = [&](int64_t row0) {
#pragma unroll
        for (int i = 0; i < (kTM * RK + 255) / 256; ++i) {
            const int e = tid + 256 * i, o = e / RK, kk = e % RK;
            breg[i] = (e < kTM * RK && row0 + o < rend && kk < r) ? B[(row0 + o) * r + kk] : 0.f;
        }
    };
    load_gm_tile<T>(G, rend, in_f, ldg, mask, rbeg, colb, vec, rg);
    load_b_tile(rbeg);
    for (int64_t row0 = rbeg; row0 < rend; row0 += kTM) {
        __syncthreads();                                                     // the previous tile has been consumed
        store_gm_tile<T>(rg, rend, in_f, sparse, scaling, ab_code, row0, colb, gm);
#pragma unroll
        for (int i = 0; i < (kTM * RK + 255) / 256; ++i) {
            const int e = tid + 256 * i;
            if (e < kTM * RK) Bs[(e / RK) * (RK + 1) + e % RK] = round_code(breg[i], ab_code);
        }
        __syncthreads();
        if (row0 + kTM < rend) {                                             // next tile's loads fly during the MFMAs
            load_gm_tile<T>(G, rend, in_f, ldg, mask, row0 + kTM, colb, vec, rg);
            load_b_tile(row0 + kTM);
        }
        if (pb) {
            // wave: rows (w&1)*16.., column half (w>>1)*128.. of the tile
            f32x4 acc_b[NT16];
#pragma unroll
            for (int n = 0; n < NT16; ++n) acc_b[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            // 8 rank-4 steps per batch: all LDS operands of a batch are requested before its first MFMA
            for (int i0 = chalf * 128; i0 < chalf * 128 + 128; i0 += 32) {
                float x[8], av[NT16][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x[u] = gm[(rh * 16 + l15) * kLd + i0 + 4 * u + q];          // Gm[o = l&15][i = i0 + 4u + (l>>4)]
#pragma unroll
                    for (int n = 0; n < NT16; ++n) av[n][u] = As[(n * 16 + l15) * kLd + i0 + 4 * u + q];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int n = 0; n < NT16; ++n) acc_b[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u], av[n][u], acc_b[n], 0, 0, 0);
            }
#pragma unroll
            for (int n = 0; n < NT16; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) red[chalf][(rh * 16 + 4 * q + g) * RK + n * 16 + l15] = acc_b[n][g];
        }
        if (part_a) {
            for (int o0 = 0; o0 < kTM; o0 += 8) {
                float y[2][4], bq[2][NT16];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[u][c] = gm[(o0 + 4 * u + q) * kLd + wave * 64 + c * 16 + l15];   // Gm[o][i]
#pragma unroll
                    for (int n = 0; n < NT16; ++n) bq[u][n] = Bs[(o0 + 4 * u + q) * (RK + 1) + n * 16 + l15];      // B^T[k][o]
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int n = 0; n < NT16; ++n)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc_a[n][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[u][n], y[u][c], acc_a[n][c], 0, 0, 0);
            }
        }
        if (pb) {
            __syncthreads();
            for (int e = tid; e < kTM * RK; e += 256) {          // two column halves, fixed order
                const int o = e / RK, kk = e % RK;
                if (row0 + o < rend && kk < r) pb[(row0 + o) * r + kk] = ieee_add(red[0][e], red[1][e]);
            }
        }
    }
    if (part_a) {
        float *dA = part_a + int64_t(blockIdx.y) * r * in_f;
#pragma unroll
        for (int n = 0; n < NT16; ++n)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kk = n * 16 + 4 * q + g;
                    const int64_t ci = colb + wave * 64 + c * 16 + l15;
                    if (kk < r && ci < in_f) dA[int64_t(kk) * in_f + ci] = acc_a[n][c][g];
                }
    }
}

// out[i] = sum over the partial slabs, in slab order (deterministic), rounded to the autocast dtype
__global__ void lora_grad_reduce_kernel(const float *__restrict__ part, int slabs, int64_t n, int ab_code,
                                        float *__restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= slabs; s += 8) {                 // 8 loads in flight, summed in slab order
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[int64_t(s + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v = ieee_add(v, t[u]);
    }
    for (; s < slabs; ++s) v = ieee_add(v, part[int64_t(s) * n + i]);
    out[i] = round_code(v, ab_code);
}

// both reductions of a gradient pass in ONE launch (two launches of ~5 us each per layer were 2 ms of a 160 ms RESSA step)
__global__ void lora_grad_reduce2_kernel(const float *__restrict__ part_a, int slabs_a, int64_t n_a, float *__restrict__ out_a,
                                         const float *__restrict__ part_b, int slabs_b, int64_t n_b, float *__restrict__ out_b,
                                         int ab_code, unsigned blocks_a) {
    const bool is_a = blockIdx.x < blocks_a;
    const float *part = is_a ? part_a : part_b;
    const int slabs = is_a ? slabs_a : slabs_b;
    const int64_t n = is_a ? n_a : n_b;
    float *out = is_a ? out_a : out_b;
    const int64_t i = int64_t(is_a ? blockIdx.x : blockIdx.x - blocks_a) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= slabs; s += 8) {                 // 8 loads in flight, summed in slab order (as lora_grad_reduce_kernel)
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[int64_t(s + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v = ieee_add(v, t[u]);
    }
    for (; s < slabs; ++s) v = ieee_add(v, part[int64_t(s) * n + i]);
    out[i] = round_code(v, ab_code);
}

// Row splits of the gradient pass: strips x splits workgroups, ALL resident at once -- two per CU (56 KB of LDS each).  (The
// first version aimed at ">= 768 = 3 per CU": 512 ran, the other 256 made a second, half-empty round -- 4096 x 4096: 68 -> 54 us,
// 11008 x 4096: 134 -> 101 us with one full round.)
static int64_t grad_row_splits(int64_t out_f, int64_t in_f) {
    static const int64_t resident = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return int64_t(2) * n;
    }();
    const int64_t strips = (in_f + kTN - 1) / kTN, tiles = (out_f + kTM - 1) / kTM;
    int64_t s = resident / strips;                   // the most splits that still fit one round
    if (s > tiles) s = tiles;
    if (s < 1) s = 1;
    return s;
}
static size_t grad_ws_a(int64_t out_f, int64_t in_f, int r) {
    return round_up(size_t(grad_row_splits(out_f, in_f)) * size_t(r) * size_t(in_f) * 4, 256);
}
static size_t grad_ws_b(int64_t out_f, int64_t in_f, int r) {
    return round_up(size_t((in_f + kTN - 1) / kTN) * size_t(out_f) * size_t(r) * 4, 256);
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
                      float scaling, const uint8_t *mask, int sparse, int ab_code, float *dA, float *dB, char *ws,
                      hipStream_t st) {
    using raw = typename T::raw;
    const raw *g = static_cast<const raw *>(G);
    const unsigned strips = unsigned((in_f + kTN - 1) / kTN);
    const int64_t splits = grad_row_splits(out_f, in_f);
    const int64_t tiles = (out_f + kTM - 1) / kTM;
    const int64_t rows_per_split = (tiles + splits - 1) / splits * kTM;
    float *part_a = dA ? reinterpret_cast<float *>(ws) : nullptr;
    float *part_b = dB ? reinterpret_cast<float *>(ws + grad_ws_a(out_f, in_f, r)) : nullptr;
#define VLMC_GRAD(NT)                                                                                                          \
    hipLaunchKernelGGL((lora_grad_fused_kernel<T, NT>), dim3(strips, unsigned(splits)), dim3(256), 0, st, g, out_f, in_f, ldg, A, \
                       B, r, scaling, mask, sparse, ab_code, rows_per_split, part_a, part_b)
    const int nt = (r + 15) / 16;
    switch (nt) {
        case 1: VLMC_GRAD(1); break;
        case 2: VLMC_GRAD(2); break;
        case 3: VLMC_GRAD(3); break;
        default: VLMC_GRAD(4); break;
    }
#undef VLMC_GRAD
    if (dA && dB) {
        const int64_t na = int64_t(r) * in_f, nb = out_f * int64_t(r);
        const unsigned ba = unsigned((na + 255) / 256), bb = unsigned((nb + 255) / 256);
        hipLaunchKernelGGL(lora_grad_reduce2_kernel, dim3(ba + bb), dim3(256), 0, st, part_a, int(splits), na, dA, part_b, int(strips), nb, dB,
                           ab_code, ba);
        VLMC_HIP_CHECK_LAUNCH("vlmc_lora_grad");
        return VLMC_OK;
    }
    if (dA) {
        const int64_t n = int64_t(r) * in_f;
        hipLaunchKernelGGL(lora_grad_reduce_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, part_a, int(splits), n,
                           ab_code, dA);
    }
    if (dB) {
        const int64_t n = out_f * int64_t(r);
        hipLaunchKernelGGL(lora_grad_reduce_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, part_b, int(strips), n,
                           ab_code, dB);
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
    return grad_ws_a(out_features, in_features, r) + grad_ws_b(out_features, in_features, r);
}

extern "C" int vlmc_lora_grad(const void *G, int dtype, int64_t out_features, int64_t in_features, int64_t ldg, const float *A,
                              const float *B, int r, float scaling, const uint8_t *mask, int sparse, int ab_code,
                              float *dA, float *dB, void *workspace, size_t workspace_bytes, void *stream) {
    VLMC_REQUIRE(G && A && B && mask, "vlmc_lora_grad: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && ldg >= in_features && r > 0 && r <= 64,
                 "vlmc_lora_grad: bad shape out=%lld in=%lld r=%d (r <= 64)", (long long)out_features, (long long)in_features, r);
    if (dA || dB) {
        const size_t need = vlmc_lora_grad_workspace(out_features, in_features, r);
        VLMC_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) % 256) == 0, "vlmc_lora_grad: workspace missing or not 256-B aligned");
        if (workspace_bytes < need) {
            set_error("vlmc_lora_grad: workspace %zu B < required %zu B", workspace_bytes, need);
            return VLMC_EWORKSPACE;
        }
    }
    hipStream_t st = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    switch (dtype) {
        case VLMC_F32: return grad_typed<f32_t>(G, out_features, in_features, ldg, A, B, r, scaling, mask, sparse, ab_code, dA, dB, ws, st);
        case VLMC_F16: return grad_typed<f16_t>(G, out_features, in_features, ldg, A, B, r, scaling, mask, sparse, ab_code, dA, dB, ws, st);
        case VLMC_BF16: return grad_typed<bf16_t>(G, out_features, in_features, ldg, A, B, r, scaling, mask, sparse, ab_code, dA, dB, ws, st);
    }
    set_error("vlmc_lora_grad: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}
