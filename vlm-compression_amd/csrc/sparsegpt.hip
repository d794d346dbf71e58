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
#include <cstdlib>

#include "common.hpp"
#include "topk_order.hpp"

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
        uint32_t pruned = 0, at_n = 0, at_n1 = 1;
#pragma unroll
        for (int a = 0; a < kMax; ++a) {
            if (a < prune_m && i + a < count) {
                int rank = 0;
#pragma unroll
                for (int b = 0; b < kMax; ++b)
                    if (b < prune_m) rank += (t[b] < t[a] || (t[b] == t[a] && b < a)) ? 1 : 0;
                pruned |= (rank < prune_n ? 1u : 0u) << a;
                at_n = rank == prune_n - 1 ? t[a] : at_n;
                at_n1 = rank == prune_n ? t[a] : at_n1;
            }
        }
        // a tie across the selection boundary (:191 `torch.topk`): the columns it returns on the CPU (topk_order.hpp).  Whole groups
        // of a compile-time m only; a run-time m keeps the stable order (documented in DESIGN.md §2).
        if (M > 2 && prune_n > 0 && prune_n < M && i + M <= count && at_n == at_n1) {
            uint32_t gk[kMax];
#pragma unroll
            for (int a = 0; a < kMax; ++a) gk[a] = t[a];
            pruned = torch_cpu_smallest<kMax>(gk, prune_n);
        }
#pragma unroll
        for (int a = 0; a < kMax; ++a) {
            const int col = i + a;
            if (a < prune_m && col < count && ((pruned >> a) & 1u) && lane == (col & 63)) {
                if (col >= 64) m1[r] = 1; else m0[r] = 1;
            }
        }
    }
}

// The same decision for m = 4 / 8 (a group never straddles the two 64-column halves) with every lane scoring ITS OWN column:
// one division per lane and group instead of m (the version above computes the m scores of the group in every lane), the
// group's keys gathered by v_readlane, and a lane ranks only its own key: m compares instead of m^2.  Same arithmetic per
// score, same (key, column) order: the same mask.
template <int kSgRows, int M>
__device__ __forceinline__ void nm_decide_lanes(int i, int count, int lane, int prune_n, float dsq0, float dsq1,
                                                const float (&w0)[kSgRows], const float (&w1)[kSgRows], int (&m0)[kSgRows],
                                                int (&m1)[kSgRows]) {
    static_assert(64 % M == 0, "a group lies inside one half");
    const bool hi = i >= 64;
    const int l0 = i & 63;
    const bool mine = lane >= l0 && lane < l0 + M;                     // this lane's column belongs to the group
    const bool incount = (hi ? 64 : 0) + lane < count;
#pragma unroll
    for (int r = 0; r < kSgRows; ++r) {
        const float wv = hi ? w1[r] : w0[r];
        const uint32_t key = incount ? score_key(ieee_div(ieee_mul(wv, wv), hi ? dsq1 : dsq0)) : 0xFFFFFFFFu;
        int rank = 0, less = 0, equal = 0;
        uint32_t gk[M];
#pragma unroll
        for (int b = 0; b < M; ++b) {
            const uint32_t tb = gk[b] = uint32_t(__builtin_amdgcn_readlane(int(key), l0 + b));
            rank += (tb < key || (tb == key && l0 + b < lane)) ? 1 : 0;
            less += tb < key ? 1 : 0;
            equal += tb == key ? 1 : 0;
        }
        bool pruned = rank < prune_n;
        // a run of equal scores that straddles the selection boundary (:191 `torch.topk`): the columns it returns on the CPU
        // (topk_order.hpp); whole groups only.  (wave-uniform: the group's lanes vote)
        const bool straddles = mine && less < prune_n && less + equal > prune_n;
        if (M > 2 && prune_n > 0 && prune_n < M && (hi ? 64 : 0) + l0 + M <= count && __ballot(straddles) != 0)
            pruned = (torch_cpu_smallest<M>(gk, prune_n) >> ((lane - l0) & (M - 1))) & 1u;
        if (mine && incount && pruned) {
            if (hi) m1[r] = 1; else m0[r] = 1;
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
        float dsq0 = 1.f, dsq1 = 1.f;                                // squared diagonal entries of this lane's two columns (n:m scores)
        if constexpr (NM > 1) {
            const float d0 = c0 ? sU[lane * kSgBlock + lane] : 1.f, d1 = c1 ? sU[(lane + 64) * kSgBlock + lane + 64] : 1.f;
            dsq0 = ieee_mul(d0, d0);
            dsq1 = ieee_mul(d1, d1);
        }
        auto decide = [&](int i) {
            if constexpr (NM > 1) {
                if (i % NM == 0) nm_decide_lanes<kSgRows, NM>(i, count, lane, prune_n, dsq0, dsq1, w0, w1, m0, m1);
            } else if constexpr (NM == 1) {
                if (i % prune_m == 0) nm_decide<kSgRows, 0>(i, count, lane, prune_n, prune_m, sU, w0, w1, m0, m1);
            }
        };
        for (int i = 0; i < half; ++i) {
            decide(i);
            const float h0 = h0n, h1 = h1n, d = dn;
            if (i + 1 < count) {
                h0n = sU[(i + 1) * kSgBlock + lane];
                h1n = sU[(i + 1) * kSgBlock + lane + 64];
                dn = sU[(i + 1) * kSgBlock + i + 1];
            }
            sweep_step<kSgRows, false>(i, lane, h0, h1, d, w0, w1, e0, e1, m0, m1);
        }
        for (int i = 64; i < count; ++i) {
            decide(i);
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


// ------------------------------------------------------------------------------------------
// Unstructured mode in ONE launch: the block threshold (sparsegpt_pruner.py:183-185) AND the column sweep.
//     tmp = W1^2 / diag(Hinv1)^2;  thresh = sort(tmp.flatten())[int(tmp.numel() * sparsity)];  mask1 = tmp <= thresh
// used to be ~9 launches per 128 columns (pow, divide, the multi-tensor radix select's passes, logical_not) in front of
// the sweep -- 0.5 s of host time per SparseGPT prune of FlanT5-XL.  Here the rows a wave sweeps are the rows whose scores
// it ranks: lanes are columns, a wave holds R rows, the keys of its 2 R scores per lane stay in registers, and the
// workgroups of the launch agree on the threshold key in three radix levels (11 / 11 / 10 key bits; per level: LDS
// histogram -> device-scope atomics into the scope's global histogram -> grid barrier -> every workgroup scans the merged
// bins itself).  Several linears that share a factor are stacked along the rows (vlmc/sparsegpt.py: fasterprune_group);
// each is a SCOPE with its own rank.  Grid barriers as in wanda_select.hip (matrix_fused_kernel): arrival counters,
// device-scope atomics only, a bounded wait whose failure is all-or-none -- then nothing has been written, and the last
// workgroup to finish (done counter) does the whole block alone (thresholds by a streaming radix select, then every row
// group).  The workspace (kSelWsWords words) must be zero on entry and is returned zero.
// ------------------------------------------------------------------------------------------
#ifndef VLMC_SEL_DBG
#define VLMC_SEL_DBG 0
#endif
constexpr int kSelScopes = 4, kSelBins = 2048, kSelLevels = 3;
constexpr int kSelHistWords = kSelScopes * kSelLevels * kSelBins;
constexpr int kSelCtl = kSelHistWords;            // [0..2] the levels' barriers, [3] done counter
constexpr int kSelWsWords = kSelHistWords + 16;
constexpr uint32_t kSelBarFail = 0x80000000u;
constexpr uint32_t kSelSpinMax = 1u << 12;        // x (device-scope load + s_sleep, ~2.2 us) ~ 9 ms

struct SelScopes {
    int n;
    int wg0[kSelScopes + 1];                      // workgroups [wg0[s], wg0[s + 1]) own the rows of scope s
    int row0[kSelScopes + 1];                     // rows [row0[s], row0[s + 1])
    uint32_t rank[kSelScopes];                    // 0-based rank of the threshold among the scope's rows x count scores
};

__device__ __forceinline__ uint32_t sel_ld_dev(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tid 0: arrive and wait for n arrivals; false = the barrier failed (for every workgroup or for none, see wanda_select.hip)
__device__ __forceinline__ bool sel_arrive_wait(uint32_t *ctr, uint32_t n) {
    uint32_t v = atomicAdd(ctr, 1u) + 1u;
    for (uint32_t it = 0;; ++it) {
        if (v & kSelBarFail) return false;
        if (v >= n) return true;
        if (it >= kSelSpinMax) {
            const uint32_t seen = atomicCAS(ctr, v, v | kSelBarFail);
            if (seen == v) return false;
            v = seen;
            continue;
        }
        __builtin_amdgcn_s_sleep(8);
        v = sel_ld_dev(ctr);
    }
}

__device__ __forceinline__ uint32_t sel_wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = uint32_t(__shfl_up(int(v), off, 64));
        if (int(threadIdx.x & 63) >= off) v += o;
    }
    return v;
}

// 256 threads, PER bins each (thread t holds bins t * PER ..): the bin with cum <= need < cum + h, and cum.
template <int PER>
__device__ __forceinline__ void sel_find_rank(const uint32_t (&h)[PER], uint32_t need, uint32_t *red, uint32_t &bin, uint32_t &before) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += h[i];
    const uint32_t incl = sel_wave_incl_scan(s);
    if (lane == 63) red[wave] = incl;
    if (tid == 0) { red[8] = 0xFFFFFFFFu; red[9] = 0; }
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += red[w];
    const uint32_t hi = off + incl, lo = hi - s;
    if (lo <= need && need < hi) {
        uint32_t cum = lo;
        int i = 0;
        for (; i < PER - 1; ++i) {
            if (cum + h[i] > need) break;
            cum += h[i];
        }
        red[8] = uint32_t(tid * PER + i);
        red[9] = cum;
    }
    __syncthreads();
    bin = red[8];
    before = red[9];
    __syncthreads();
}

__device__ __forceinline__ void stage_factor_block(float *sU, const float *__restrict__ U1, int64_t ldu, int count) {
    const int tid = threadIdx.x;
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

__device__ __forceinline__ uint32_t sel_score_key(float w, float dsq) { return score_key(ieee_div(ieee_mul(w, w), dsq)); }   // (:183)

// the rows [r0, r0 + R) of one wave: mask from the threshold key, column sweep, stores
template <int R>
__device__ __forceinline__ void select_sweep_rows(float *__restrict__ W, int64_t r0, int64_t row_end, int count, int64_t ldw,
                                                  const float *sU, uint32_t thrkey, float *__restrict__ Err1, int64_t lde,
                                                  uint8_t *__restrict__ mask_out, int64_t ldmo) {
    const int lane = threadIdx.x & 63;
    const bool c0 = lane < count, c1 = lane + 64 < count;
    const float d0 = c0 ? sU[lane * kSgBlock + lane] : 1.f, d1 = c1 ? sU[(lane + 64) * kSgBlock + lane + 64] : 1.f;
    const float q0 = ieee_mul(d0, d0), q1 = ieee_mul(d1, d1);
    const bool any = thrkey != 0xFFFFFFFFu;                      // a NaN threshold: `tmp <= thresh` holds nowhere
    float w0[R], w1[R], e0[R], e1[R];
    int m0[R], m1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = r0 + r;
        const bool live = row < row_end;
        w0[r] = (live && c0) ? W[row * ldw + lane] : 0.f;
        w1[r] = (live && c1) ? W[row * ldw + lane + 64] : 0.f;
        m0[r] = (live && c0 && any && sel_score_key(w0[r], q0) <= thrkey) ? 1 : 0;
        m1[r] = (live && c1 && any && sel_score_key(w1[r], q1) <= thrkey) ? 1 : 0;
        e0[r] = e1[r] = 0.f;
    }
    float h0n = sU[lane], h1n = sU[lane + 64], dn = sU[0];
    const int half = count < 64 ? count : 64;
    for (int i = 0; i < half; ++i) {
        const float h0 = h0n, h1 = h1n, d = dn;
        if (i + 1 < count) {
            h0n = sU[(i + 1) * kSgBlock + lane];
            h1n = sU[(i + 1) * kSgBlock + lane + 64];
            dn = sU[(i + 1) * kSgBlock + i + 1];
        }
        sweep_step<R, false>(i, lane, h0, h1, d, w0, w1, e0, e1, m0, m1);
    }
    for (int i = 64; i < count; ++i) {
        const float h1 = h1n, d = dn;
        if (i + 1 < count) {
            h1n = sU[(i + 1) * kSgBlock + lane + 64];
            dn = sU[(i + 1) * kSgBlock + i + 1];
        }
        sweep_step<R, true>(i, lane, 0.f, h1, d, w0, w1, e0, e1, m0, m1);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = r0 + r;
        if (row >= row_end) continue;
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

// the threshold key of one scope by ONE workgroup, streaming over W (the failed launch's last workgroup)
__device__ uint32_t sel_scope_threshold_alone(const float *__restrict__ W, int64_t row_a, int64_t row_b, int count, int64_t ldw,
                                              const float *sU, uint32_t need, uint32_t *lh, uint32_t *red) {
    const int tid = threadIdx.x;
    const int64_t total = (row_b - row_a) * count;
    uint32_t prefix = 0, pmask = 0;
    for (int sh = 24; sh >= 0; sh -= 8) {
        lh[tid] = 0;
        __syncthreads();
        for (int64_t e = tid; e < total; e += 256) {
            const int64_t row = row_a + e / count;
            const int col = int(e % count);
            const float d = sU[col * kSgBlock + col];
            const uint32_t key = sel_score_key(W[row * ldw + col], ieee_mul(d, d));
            if ((key & pmask) == prefix) atomicAdd(&lh[(key >> sh) & 255u], 1u);
        }
        __syncthreads();
        uint32_t h[1] = {lh[tid]};
        uint32_t bin, before;
        sel_find_rank<1>(h, need, red, bin, before);
        prefix |= bin << sh;
        pmask |= 0xFFu << sh;
        need -= before;
    }
    return prefix;
}

template <int R>
__global__ __launch_bounds__(256) void sgpt_select_sweep_kernel(float *__restrict__ W, int count, int64_t ldw,
                                                                const float *__restrict__ U1, int64_t ldu, const SelScopes sc,
                                                                float *__restrict__ Err1, int64_t lde, uint8_t *__restrict__ mask_out,
                                                                int64_t ldmo, uint32_t *__restrict__ ws, int force_fail) {
    extern __shared__ __attribute__((aligned(16))) float sU[];   // [count][kSgBlock]
    __shared__ uint32_t lh[kSelBins];
    __shared__ uint32_t red[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    stage_factor_block(sU, U1, ldu, count);
    __syncthreads();
    int s = 0;
    while (s + 1 < sc.n && int(blockIdx.x) >= sc.wg0[s + 1]) ++s;
    const int64_t row_end = sc.row0[s + 1];
    const int64_t r0 = sc.row0[s] + (int64_t(int(blockIdx.x) - sc.wg0[s]) * 4 + wave) * R;
    const bool c0 = lane < count, c1 = lane + 64 < count;
    // ---- the keys of my rows' scores ------------------------------------------------------------------------------
    uint32_t k0[R], k1[R];
    uint32_t valid = 0;                                          // bit 2 r: (row r, lane), bit 2 r + 1: (row r, lane + 64)
    {
        const float d0 = c0 ? sU[lane * kSgBlock + lane] : 1.f, d1 = c1 ? sU[(lane + 64) * kSgBlock + lane + 64] : 1.f;
        const float q0 = ieee_mul(d0, d0), q1 = ieee_mul(d1, d1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = r0 + r;
            const bool live = row < row_end;
            k0[r] = k1[r] = 0;
            if (live && c0) { k0[r] = sel_score_key(W[row * ldw + lane], q0); valid |= 1u << (2 * r); }
            if (live && c1) { k1[r] = sel_score_key(W[row * ldw + lane + 64], q1); valid |= 2u << (2 * r); }
        }
    }
    // ---- three radix levels ------------------------------------------------------------------------------------------
    uint32_t prefix = 0, pmask = 0, need = sc.rank[s];
    bool fail = false;
#pragma unroll 1
    for (int lvl = 0; lvl < kSelLevels; ++lvl) {
        const int shift = lvl == 0 ? 21 : (lvl == 1 ? 10 : 0);
        const uint32_t bmask = lvl == 2 ? 1023u : 2047u;
        for (int i = tid; i < kSelBins; i += 256) lh[i] = 0;
        __syncthreads();
#if !(VLMC_SEL_DBG & 1)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (((valid >> (2 * r)) & 1u) && (k0[r] & pmask) == prefix) atomicAdd(&lh[(k0[r] >> shift) & bmask], 1u);
            if (((valid >> (2 * r + 1)) & 1u) && (k1[r] & pmask) == prefix) atomicAdd(&lh[(k1[r] >> shift) & bmask], 1u);
        }
#endif
        __syncthreads();
        uint32_t *gh = ws + (s * kSelLevels + lvl) * kSelBins;
#if !(VLMC_SEL_DBG & 2)
        for (int i = tid; i < kSelBins; i += 256) {
            const uint32_t v = lh[i];
            if (v) atomicAdd(&gh[i], v);
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my atomics have been performed
        __syncthreads();
#if !(VLMC_SEL_DBG & 4)
        if (tid == 0) red[10] = sel_arrive_wait(ws + kSelCtl + lvl, gridDim.x) ? 1u : 0u;
#else
        if (tid == 0) red[10] = 1;
#endif
        __syncthreads();
        fail = red[10] == 0 || force_fail == lvl + 1;            // (test hook: every workgroup sees the same value)
        if (fail) break;
        uint32_t h[8];
#pragma unroll
#if !(VLMC_SEL_DBG & 8)
        for (int i = 0; i < 8; ++i) h[i] = sel_ld_dev(&gh[tid * 8 + i]);
#else
        for (int i = 0; i < 8; ++i) h[i] = lh[tid * 8 + i] + (tid == 0 && i == 0 ? 0x7FFFFFFFu : 0u);
#endif
        uint32_t bin, before;
        sel_find_rank<8>(h, need, red, bin, before);
        prefix |= bin << shift;
        pmask |= bmask << shift;
        need -= before;
    }
    // my last look at the workspace is behind me
    if (tid == 0) red[11] = atomicAdd(&ws[kSelCtl + 3], 1u);
    if (!fail) select_sweep_rows<R>(W, r0, row_end, count, ldw, sU, prefix, Err1, lde, mask_out, ldmo);
    __syncthreads();
    if (red[11] != gridDim.x - 1) return;
    // ---- the launch's last workgroup: a failed launch is done here, alone; then the workspace goes back to zero ----
    if (fail) {
        for (int q = 0; q < sc.n; ++q) {
            const uint32_t thr = sel_scope_threshold_alone(W, sc.row0[q], sc.row0[q + 1], count, ldw, sU, sc.rank[q], lh, red);
            for (int64_t g0 = sc.row0[q] + int64_t(wave) * R; g0 < sc.row0[q + 1]; g0 += 4 * R)
                select_sweep_rows<R>(W, g0, sc.row0[q + 1], count, ldw, sU, thr, Err1, lde, mask_out, ldmo);
            __syncthreads();
        }
    }
    for (int i = tid; i < kSelWsWords; i += 256) __hip_atomic_store(&ws[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

extern "C" int64_t vlmc_sparsegpt_select_workspace_bytes(void) { return int64_t(kSelWsWords) * 4; }

extern "C" int vlmc_sparsegpt_select_sweep(float *W, int64_t count, int64_t ldw, const float *U1, int64_t ldu, int n_scopes,
                                           const int64_t *scope_rows, const int64_t *scope_ranks, float *Err1, int64_t lde,
                                           uint8_t *mask_out, int64_t ldmo, void *workspace, void *stream) {
    VLMC_REQUIRE(W && U1 && Err1 && workspace && scope_rows && scope_ranks, "vlmc_sparsegpt_select_sweep: null pointer");
    VLMC_REQUIRE(count > 0 && count <= kSgBlock && ldw >= count && ldu >= count && lde >= count,
                 "vlmc_sparsegpt_select_sweep: bad shape count=%lld (max %d columns per block)", (long long)count, kSgBlock);
    VLMC_REQUIRE(n_scopes >= 1 && n_scopes <= kSelScopes, "vlmc_sparsegpt_select_sweep: 1..%d scopes, got %d", kSelScopes, n_scopes);
    int64_t total = 0;
    for (int q = 0; q < n_scopes; ++q) {
        VLMC_REQUIRE(scope_rows[q] > 0 && scope_rows[q] * count < (int64_t(1) << 31),
                     "vlmc_sparsegpt_select_sweep: scope %d has %lld rows", q, (long long)scope_rows[q]);
        VLMC_REQUIRE(scope_ranks[q] >= 0 && scope_ranks[q] < scope_rows[q] * count,
                     "vlmc_sparsegpt_select_sweep: rank %lld outside scope %d", (long long)scope_ranks[q], q);
        total += scope_rows[q];
    }
    // every workgroup of the launch must be resident at once (grid barriers): two workgroups per CU (72 KB of LDS each)
    int dev = 0;
    static int cus[64] = {0};
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 64 && cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 64;
        cus[dev] = v;
    }
    const int64_t max_wgs = 2 * int64_t(dev < 64 ? cus[dev] : 64);
    int rows = 0;
    for (int r : {1, 2, 4, 8}) {
        int64_t wgs = 0;
        for (int q = 0; q < n_scopes; ++q) wgs += (scope_rows[q] + 4 * r - 1) / (4 * r);
        if (wgs <= max_wgs) { rows = r; break; }
    }
    VLMC_REQUIRE(rows != 0, "vlmc_sparsegpt_select_sweep: %lld rows do not fit %lld co-resident workgroups of 4 x {1,2,4,8} rows",
                 (long long)total, (long long)max_wgs);
    SelScopes sc;
    sc.n = n_scopes;
    sc.wg0[0] = 0;
    sc.row0[0] = 0;
    for (int q = 0; q < n_scopes; ++q) {
        sc.wg0[q + 1] = sc.wg0[q] + int((scope_rows[q] + 4 * rows - 1) / (4 * rows));
        sc.row0[q + 1] = sc.row0[q] + int(scope_rows[q]);
        sc.rank[q] = uint32_t(scope_ranks[q]);
    }
    for (int q = n_scopes; q < kSelScopes; ++q) { sc.wg0[q + 1] = sc.wg0[q]; sc.row0[q + 1] = sc.row0[q]; sc.rank[q] = 0; }
    const size_t lds = size_t(count) * kSgBlock * sizeof(float);
    static PerDeviceOnce once;
    int d2;
    if (once.needed(&d2)) {
        const int bytes = kSgBlock * kSgBlock * int(sizeof(float));
        bool ok = true;
#define VLMC_SELSWEEP_ATTR(R)                                                                                              \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(sgpt_select_sweep_kernel<R>),                             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
        VLMC_SELSWEEP_ATTR(1); VLMC_SELSWEEP_ATTR(2); VLMC_SELSWEEP_ATTR(4); VLMC_SELSWEEP_ATTR(8);
#undef VLMC_SELSWEEP_ATTR
        if (!ok) {
            set_error("vlmc_sparsegpt_select_sweep: cannot reserve 64 KB of LDS");
            return VLMC_EHIP;
        }
        once.mark(d2);
    }
    int force_fail = 0;
    if (const char *e = getenv("VLMC_SGPT_SELECT_FORCE_FAIL")) force_fail = atoi(e);       // tests: 1..3 = fail at that level
#define VLMC_SELSWEEP(R)                                                                                                      \
    hipLaunchKernelGGL((sgpt_select_sweep_kernel<R>), dim3(unsigned(sc.wg0[n_scopes])), dim3(256), lds, as_stream(stream), W,   \
                       int(count), ldw, U1, ldu, sc, Err1, lde, mask_out, ldmo, static_cast<uint32_t *>(workspace), force_fail)
    switch (rows) {
        case 1: VLMC_SELSWEEP(1); break;
        case 2: VLMC_SELSWEEP(2); break;
        case 4: VLMC_SELSWEEP(4); break;
        default: VLMC_SELSWEEP(8); break;
    }
#undef VLMC_SELSWEEP
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparsegpt_select_sweep");
    return VLMC_OK;
}

// ---- K10: the trailing update of a 128-column block, `W[:, i2:] -= Err1.matmul(Hinv[i1:i2, i2:])` (sparsegpt_pruner.py:210) -------
//   W[r, c] -= sum_{k < count} Err[r, k] * U[k, c]          fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32
// (rounds 2-5 handed this to the GEMM library through `torch.addmm_`).  K is the block's 128 columns.  A workgroup owns a 128 x 128
// tile of W; each of its four waves 64 x 64 of it (2 x 2 MFMA tiles, k ascending in pairs).  K goes through LDS in four chunks of 32,
// double buffered: the next chunk's 16-byte global loads are in flight during this chunk's 64 MFMAs per wave, and 68 KB of LDS let a
// second workgroup share the CU (its loads and stores run beside this one's products).  The tile of W is requested before the
// products and subtracted after them: a lane's 32 columns of a row are one 128-byte segment per half wave.  An element's arithmetic
// -- one accumulator over the block's k, one subtraction -- does not depend on which columns share the launch.  MFMA-bound by design (157 TFLOP/s fp32 matrix peak): a tile is 4.2 MFLOP = 6.8 us of a CU's matrix pipes.
namespace vlmc {
typedef float tu_f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kTuTile = 128, kTuK = 32, kTuLdE = kTuK + 1, kTuLdU = kTuTile + 4;     // Err rows are read down a column of k (odd pitch), U rows along c
constexpr int kTuBuf = kTuTile * kTuLdE + kTuK * kTuLdU;                             // floats per LDS buffer (Err chunk | U chunk)

__global__ __launch_bounds__(256, 2) void sgpt_trailing_kernel(float *__restrict__ W, int rows, int ncols, int64_t ldw,
                                                               const float *__restrict__ Err, int64_t lde, const float *__restrict__ U,
                                                               int64_t ldu, int count) {
    extern __shared__ float tu_sh[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.y * kTuTile, c0 = blockIdx.x * kTuTile;
    const bool evec = (lde & 3) == 0 && (reinterpret_cast<uintptr_t>(Err) & 15u) == 0;
    const bool uvec = (ldu & 3) == 0 && (reinterpret_cast<uintptr_t>(U) & 15u) == 0;
    // chunk q of K: thread t stages Err rows (t >> 3) + 32 i, k = 4 (t & 7) ..; U rows k = (t >> 5) + 8 i, columns 4 (t & 31) ..
    f32x4_t se[4], su[4];
    auto fetch = [&](int q) {
        const int k0 = q * kTuK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + (tid >> 3) + 32 * i, k = k0 + 4 * (tid & 7);
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (row < rows && k < count) {
                const float *src = Err + int64_t(row) * lde + k;
                if (evec && k + 3 < count) v = *reinterpret_cast<const f32x4_t *>(src);
                else
                    for (int t = 0; t < 4 && k + t < count; ++t) v[t] = src[t];
            }
            se[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + (tid >> 5) + 8 * i, c = c0 + 4 * (tid & 31);
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
            if (k < count && c < ncols) {
                const float *src = U + int64_t(k) * ldu + c;
                if (uvec && c + 3 < ncols) v = *reinterpret_cast<const f32x4_t *>(src);
                else
                    for (int t = 0; t < 4 && c + t < ncols; ++t) v[t] = src[t];
            }
            su[i] = v;
        }
    };
    auto stash = [&](int buf) {
        float *ea = tu_sh + buf * kTuBuf, *ub = ea + kTuTile * kTuLdE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = ea + ((tid >> 3) + 32 * i) * kTuLdE + 4 * (tid & 7);
            d[0] = se[i][0], d[1] = se[i][1], d[2] = se[i][2], d[3] = se[i][3];
            *reinterpret_cast<f32x4_t *>(ub + ((tid >> 5) + 8 * i) * kTuLdU + 4 * (tid & 31)) = su[i];
        }
    };
    const int rb = wave >> 1, cb = wave & 1;
    tu_f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nq = (count + kTuK - 1) / kTuK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) fetch(q + 1);                                         // in flight during this chunk's products
        const float *ea = tu_sh + (q & 1) * kTuBuf, *ub = ea + kTuTile * kTuLdE;
        const float *ap0 = ea + (rb * 64 + (lane & 31)) * kTuLdE + (lane >> 5), *ap1 = ap0 + 32 * kTuLdE;
        const float *bp0 = ub + (lane >> 5) * kTuLdU + cb * 64 + (lane & 31), *bp1 = bp0 + 32;
#pragma unroll
        for (int k = 0; k < kTuK; k += 2) {
            const float a0 = ap0[k], a1 = ap1[k], b0 = bp0[k * kTuLdU], b1 = bp1[k * kTuLdU];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (q + 1 < nq) {
            stash((q + 1) & 1);                                               // (the buffer chunk q - 1 was read from: every wave left it before the last barrier)
            __syncthreads();
        }
    }
    // ---- W -= acc: register r of a tile is row 8 (r / 4) + 4 (lane / 32) + r % 4, column lane % 32; 16 loads in flight, then 16 stores ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = c0 + cb * 64 + j * 32 + (lane & 31);
            if (col >= ncols) continue;
            const int rbase = r0 + rb * 64 + i * 32 + 4 * (lane >> 5);
            float w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + 8 * (r >> 2) + (r & 3);
                w[r] = row < rows ? W[int64_t(row) * ldw + col] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + 8 * (r >> 2) + (r & 3);
                if (row < rows) W[int64_t(row) * ldw + col] = ieee_add(w[r], -acc[i][j][r]);
            }
        }
}
}  // namespace vlmc

extern "C" int vlmc_sparsegpt_trailing_update(float *W, int64_t out_features, int64_t ncols, int64_t ldw, const float *Err1, int64_t lde,
                                              const float *U, int64_t ldu, int64_t count, void *stream) {
    VLMC_REQUIRE(W && Err1 && U, "vlmc_sparsegpt_trailing_update: null pointer");
    VLMC_REQUIRE(out_features > 0 && ncols >= 0 && count > 0 && count <= kTuTile && out_features < (int64_t(1) << 24) && ncols < (int64_t(1) << 24),
                 "vlmc_sparsegpt_trailing_update: bad shape (1 <= count <= 128)");
    VLMC_REQUIRE(ldw >= ncols && lde >= count && ldu >= ncols, "vlmc_sparsegpt_trailing_update: a row stride is shorter than its row");
    if (ncols == 0) return VLMC_OK;
    const size_t lds = size_t(2) * kTuBuf * sizeof(float);
    static PerDeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sgpt_trailing_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) !=
            hipSuccess) {
            set_error("vlmc_sparsegpt_trailing_update: cannot reserve %zu bytes of LDS", lds);
            return VLMC_EHIP;
        }
        once.mark(dev);
    }
    const dim3 grid{unsigned((ncols + kTuTile - 1) / kTuTile), unsigned((out_features + kTuTile - 1) / kTuTile)};
    hipLaunchKernelGGL(sgpt_trailing_kernel, grid, dim3(256), lds, as_stream(stream), W, int(out_features), int(ncols), ldw, Err1, lde, U,
                       ldu, int(count));
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparsegpt_trailing_update");
    return VLMC_OK;
}

// ---- the whole 128-column block loop of `fasterprune` (sparsegpt_pruner.py:167-212) issued from ONE call -------------------------------
// Per block the reference runs the threshold, a Python loop over the block's columns and the trailing product; here a block is one sweep
// launch (vlmc_sparsegpt_sweep in n:m mode, vlmc_sparsegpt_select_sweep otherwise) and one trailing-update launch, all blocks issued from
// this one call: from Python a block was ~30 us of host time (ctypes) against ~41 us of sweep on the GPU, on a rank that holds 1 / 8 of
// the samples the host paced the loop.
// Measured and NOT kept (profiles/r06_sparsegpt.md): a LOOK-AHEAD form -- the next block's 128 columns updated first, the rest of the
// trailing update on a side stream beside the next sweep (same bits: an element's arithmetic does not depend on the split).  It is 5 %
// SLOWER on configs[2] (2.02 against 1.92 s): the sweeps of a block's independent linears already run side by side on streams of their own,
// and MFMA waves sharing a SIMD with a sweep delay its chain of 128 dependent column steps by more than the overlap returns.
extern "C" int vlmc_sparsegpt_prune_blocks(float *W, int64_t out_features, int64_t in_features, int64_t ldw, const float *U, int64_t ldu,
                                           int64_t blocksize, int prune_n, int prune_m, int n_scopes, const int64_t *scope_rows,
                                           const double *scope_sparsity, float *err, int64_t lde, uint8_t *mask_out, int64_t ldmo,
                                           void *select_workspace, void *stream) {
    VLMC_REQUIRE(W && U && err, "vlmc_sparsegpt_prune_blocks: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && blocksize > 0 && blocksize <= kSgBlock && ldw >= in_features && ldu >= in_features &&
                     lde >= (blocksize < in_features ? blocksize : in_features),
                 "vlmc_sparsegpt_prune_blocks: bad shape (blocksize <= %d)", kSgBlock);
    VLMC_REQUIRE(prune_n != 0 || (select_workspace && scope_rows && scope_sparsity && n_scopes >= 1 && n_scopes <= kSelScopes),
                 "vlmc_sparsegpt_prune_blocks: the unstructured mode needs 1..%d scopes, their sparsities and the select workspace", kSelScopes);
    for (int64_t i1 = 0; i1 < in_features; i1 += blocksize) {
        const int64_t i2 = i1 + blocksize < in_features ? i1 + blocksize : in_features, count = i2 - i1;
        int rc;
        if (prune_n != 0) {
            rc = vlmc_sparsegpt_sweep(W + i1, out_features, count, ldw, U + i1 * ldu + i1, ldu, nullptr, 0, prune_n, prune_m, err, lde,
                                      mask_out ? mask_out + i1 : nullptr, ldmo, stream);
        } else {
            int64_t ranks[kSelScopes];
            for (int q = 0; q < n_scopes; ++q) {
                const int64_t numel = scope_rows[q] * count;
                const int64_t r = int64_t(double(numel) * scope_sparsity[q]);        // int(rows * (i2 - i1) * sparsity), as the reference's index (:184)
                ranks[q] = r < numel - 1 ? (r < 0 ? 0 : r) : numel - 1;
            }
            rc = vlmc_sparsegpt_select_sweep(W + i1, count, ldw, U + i1 * ldu + i1, ldu, n_scopes, scope_rows, ranks, err, lde,
                                             mask_out ? mask_out + i1 : nullptr, ldmo, select_workspace, stream);
        }
        if (rc != VLMC_OK) return rc;
        if (i2 < in_features) {
            rc = vlmc_sparsegpt_trailing_update(W + i2, out_features, in_features - i2, ldw, err, lde, U + i1 * ldu + i2, ldu, count, stream);
            if (rc != VLMC_OK) return rc;
        }
    }
    return VLMC_OK;
}
