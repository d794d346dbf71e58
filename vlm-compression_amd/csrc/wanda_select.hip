// K2-K7: fused Wanda score + mask selection + apply for one linear
// (replaces /root/reference/lavis/compression/pruners/wanda_pruner.py:318-341 and :666-687).
//
// HBM-bound: per weight element the algorithm must read W (2 B), write the bool mask (1 B) and
// write the zeroed W back (2 B) => 5 B/weight (3 B under lora_model=True).  The fp32 score
// |W|*sqrt(s) is never materialised; it lives in registers as an order-preserving u32 key.
//
//  SEL_ROW    one workgroup (NW waves, usually ONE wave => no barriers) per row; the row's keys
//             stay in VGPRs (8*CH per lane, lanes own 16-byte column chunks so loads/stores are
//             coalesced).  The k-th smallest key is found by bisection on the key bits with
//             wave-wide counting (DPP reduction), switching to an exact all-pairs rank over the
//             few keys left in the bracket; ties are broken by column index exactly like the
//             reference's stable sort.  A persistent grid walks the rows: each wave seeds the
//             bracket of its next row with the threshold of its previous one (rows of one
//             matrix share the column scales, so thresholds differ by a few percent) and
//             prefetches the next row while it works on the current one.
//  SEL_MATRIX three global radix-histogram passes (12+10+10 key bits, LDS histograms flushed with
//             integer atomics), then an elementwise apply pass.  W (<= 17 MB for ViT-g) is
//             re-read from L2/Infinity Cache, not HBM.
//  SEL_NM     elementwise: each lane ranks the columns of its m-groups in registers.
#include <cstdlib>

#include "common.hpp"

namespace vlmc {

// ------------------------------------------------------------------------------------------
// wave primitives
// ------------------------------------------------------------------------------------------
// Sum over the 64 lanes with DPP (no LDS traffic); result is wave-uniform.
__device__ __forceinline__ uint32_t wave_sum_u32_dpp(uint32_t v) {
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x140, 0xF, 0xF, true));  // row_mirror
    // every lane of a 16-lane row now holds its row's sum
    return uint32_t(__builtin_amdgcn_readlane(int(v), 0)) + uint32_t(__builtin_amdgcn_readlane(int(v), 16)) +
           uint32_t(__builtin_amdgcn_readlane(int(v), 32)) + uint32_t(__builtin_amdgcn_readlane(int(v), 48));
}

// f32 sum over the 64 lanes with DPP; every lane of the result row holds the total of its 16-lane
// row, the four row totals are combined through readlane (fixed order => deterministic).
__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for
// outstanding global loads (vmcnt), so the next row's prefetch stays in flight across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// inclusive prefix sum over lanes (rare tie path; shuffles are fine)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(v, off, kWave);
        if (lane >= off) v += o;
    }
    return v;
}

// ------------------------------------------------------------------------------------------
// chunk access: 8 consecutive columns per lane
// ------------------------------------------------------------------------------------------
template <typename T, bool ALIGNED>
__device__ __forceinline__ Chunk8<T> load_row_chunk(const typename T::raw *row, int64_t col0, int64_t in_f) {
    if constexpr (ALIGNED) {
        return load_chunk8<T>(row + col0);
    } else {
        Chunk8<T> c;
#pragma unroll
        for (int j = 0; j < 8; ++j) c.v[j] = (col0 + j < in_f) ? row[col0 + j] : typename T::raw(0);
        return c;
    }
}
template <typename T, bool ALIGNED>
__device__ __forceinline__ void store_row_chunk(typename T::raw *row, int64_t col0, int64_t in_f, const Chunk8<T> &c) {
    if constexpr (ALIGNED) {
        store_chunk8<T>(row + col0, c);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (col0 + j < in_f) row[col0 + j] = c.v[j];
    }
}
template <bool ALIGNED>
__device__ __forceinline__ void store_mask_chunk(uint8_t *mrow, int64_t col0, int64_t in_f, uint32_t keepbits) {
    if constexpr (ALIGNED) {
        // spread bit j to byte j
        uint64_t m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) m |= uint64_t((keepbits >> j) & 1u) << (8 * j);
        *reinterpret_cast<uint64_t *>(mrow + col0) = m;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (col0 + j < in_f) mrow[col0 + j] = uint8_t((keepbits >> j) & 1u);
    }
}
template <bool ALIGNED>
__device__ __forceinline__ void load_sq_chunk(const float *sq, int64_t col0, int64_t in_f, float *o) {
    if constexpr (ALIGNED) {
        float4 a = reinterpret_cast<const float4 *>(sq + col0)[0], b = reinterpret_cast<const float4 *>(sq + col0)[1];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (col0 + j < in_f) ? sq[col0 + j] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// SEL_ROW
// ------------------------------------------------------------------------------------------
constexpr int kCap = 32;   // bracket population at which bisection hands over to the exact all-pairs rank

template <int NW> struct RowSmem {
    uint32_t cnt[2][NW];       // double-buffered per-wave counts
    uint32_t cnt2[2][NW];
    uint32_t scan[NW];
    double dsum[NW];
    unsigned long long seg[NW * kCap];   // per-wave candidate segments
    unsigned long long cand[kCap];       // dense candidate list
    unsigned long long cut;
    uint32_t segcnt[NW];
};

#ifndef VLMC_COUNT_MODE
#define VLMC_COUNT_MODE 0
#endif
// wave-wide count(key <= mid).  Three codegen variants (VLMC_COUNT_MODE) for tuning:
//  0: per-lane counters (v_cmp + v_addc) reduced with DPP
//  1: ballot + scalar popcount (v_cmp -> SGPR pair, s_bcnt1, s_add): no cross-lane reduce
//  2: half of the keys each way, so the vector and the scalar unit share the work
template <int E> __device__ __forceinline__ uint32_t wave_count_le(const uint32_t (&key)[E], uint32_t mid) {
#if VLMC_COUNT_MODE == 1
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < E; ++i) c += uint32_t(__popcll(__ballot(key[i] <= mid)));
    return c;
#elif VLMC_COUNT_MODE == 2
    constexpr int H = (E * 3) / 8;   // share counted on the scalar unit
    uint32_t cs = 0, c0 = 0, c1 = 0;
#pragma unroll
    for (int i = 0; i < H; ++i) cs += uint32_t(__popcll(__ballot(key[i] <= mid)));
#pragma unroll
    for (int i = H; i + 1 < E; i += 2) {
        c0 += (key[i] <= mid) ? 1u : 0u;
        c1 += (key[i + 1] <= mid) ? 1u : 0u;
    }
    if ((E - H) & 1) c0 += (key[E - 1] <= mid) ? 1u : 0u;
    return cs + wave_sum_u32_dpp(c0 + c1);
#else
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < E; ++i) c += (key[i] <= mid) ? 1u : 0u;
    return wave_sum_u32_dpp(c);
#endif
}

template <int E, int NW>
__device__ __forceinline__ uint32_t block_count_le(const uint32_t (&key)[E], uint32_t mid, RowSmem<NW> &sm, int &phase) {
    uint32_t c = wave_count_le<E>(key, mid);
    if constexpr (NW > 1) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) sm.cnt[phase][wave] = c;
        lds_barrier();
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += sm.cnt[phase][w];
        phase ^= 1;
        c = t;
    }
    return c;
}

// count(key <= a) and count(key <= b) in one sweep (bracket verification)
template <int E, int NW>
__device__ __forceinline__ void block_count_le2(const uint32_t (&key)[E], uint32_t a, uint32_t b, RowSmem<NW> &sm,
                                                int &phase, uint32_t &ca, uint32_t &cb) {
    uint32_t x = wave_count_le<E>(key, a);
    uint32_t y = wave_count_le<E>(key, b);
    if constexpr (NW > 1) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sm.cnt[phase][wave] = x; sm.cnt2[phase][wave] = y; }
        lds_barrier();
        uint32_t tx = 0, ty = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { tx += sm.cnt[phase][w]; ty += sm.cnt2[phase][w]; }
        phase ^= 1;
        x = tx; y = ty;
    }
    ca = x; cb = y;
}

template <typename T, int CH, int NW, bool ALIGNED>
__global__ __launch_bounds__(64 * NW, 4) void select_rows_kernel(typename T::raw *__restrict__ W, int64_t out_f,
                                                              int64_t in_f, int64_t ldw,
                                                              const float *__restrict__ sqrt_scaler, uint32_t k,
                                                              int apply_zero, uint8_t *__restrict__ mask,
                                                              double *__restrict__ row_sums) {
    constexpr int NT = 64 * NW;
    constexpr int E = CH * 8;
    __shared__ RowSmem<NW> sm;
    const int tid = threadIdx.x;
    const int64_t nchunks = (in_f + 7) / 8;

    // per-column sqrt(scaler_row) (wanda_pruner.py:318), loaded once per workgroup and kept in
    // registers for every row it handles
    float sqv[E];
    bool valid[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        const int64_t c = int64_t(s) * NT + tid;
        valid[s] = c < nchunks;
        if (valid[s]) {
            load_sq_chunk<ALIGNED>(sqrt_scaler, c * 8, in_f, &sqv[s * 8]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) sqv[s * 8 + j] = 0.f;
        }
    }

    // software prefetch: the next row's chunks are in flight while the current row is processed
    Chunk8<T> nxt[CH];
    int64_t row = blockIdx.x;
    if (row < out_f) {
#pragma unroll
        for (int s = 0; s < CH; ++s)
            if (valid[s]) nxt[s] = load_row_chunk<T, ALIGNED>(W + row * ldw, (int64_t(s) * NT + tid) * 8, in_f);
    }
    bool have_prev = false;
    uint32_t prev_key = 0;

    for (; row < out_f; row += gridDim.x) {
        typename T::raw *wrow = W + row * ldw;
        Chunk8<T> raw[CH];
#pragma unroll
        for (int s = 0; s < CH; ++s) raw[s] = nxt[s];
        const int64_t nrow = row + gridDim.x;
        if (nrow < out_f) {
#pragma unroll
            for (int s = 0; s < CH; ++s)
                if (valid[s]) nxt[s] = load_row_chunk<T, ALIGNED>(W + nrow * ldw, (int64_t(s) * NT + tid) * 8, in_f);
        }
        uint32_t key[E];
        float fsum = 0.f;
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int64_t col0 = (int64_t(s) * NT + tid) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = valid[s] && (ALIGNED || col0 + j < in_f);
                if (live) {
                    const float sc = ieee_mul(fabsf(to_f32<T>(raw[s].v[j])), sqv[s * 8 + j]);
                    key[s * 8 + j] = score_key(sc);
                    fsum += sc;
                } else {
                    key[s * 8 + j] = 0xFFFFFFFFu;   // padding sorts after every real column
                }
            }
        }

        // column of this lane's first element; opaque to the optimiser so that the 8*CH per-key
        // column indices derived from it are not hoisted out of the row loop into registers
        uint32_t tid8 = uint32_t(tid) * 8u;
        asm volatile("" : "+v"(tid8));

        // ---- find the cut: the (key, column) pair of rank k-1 in stable order ----------
        unsigned long long cut = 0;   // prune (key,col) <= cut
        if (k > 0) {
            int phase = 0;
            uint32_t lo = 0, hi = 0xFFFFFFFFu;
            uint32_t cb = 0;                    // count(key <  lo)
            uint32_t ca = uint32_t(NT) * E;     // count(key <= hi)
            if (have_prev) {
                // guess: this row's threshold lies within +-2^20 key units (6..12 % in value) of the
                // previous row's.  Verified by counting, so a wrong guess only costs time.
                constexpr uint32_t D = 1u << 20;
                const uint32_t glo = prev_key > D ? prev_key - D : 0u;
                const uint32_t ghi = prev_key < 0xFFFFFFFFu - D ? prev_key + D : 0xFFFFFFFFu;
                uint32_t c_below = 0, c_upto = 0;   // count(key < glo), count(key <= ghi)
                if (glo > 0) {
                    block_count_le2<E, NW>(key, glo - 1, ghi, sm, phase, c_below, c_upto);
                } else {
                    c_upto = block_count_le<E, NW>(key, ghi, sm, phase);
                }
                if (c_below >= k) { hi = glo - 1; ca = c_below; }
                else if (c_upto < k) { lo = ghi + 1; cb = c_upto; }
                else { lo = glo; hi = ghi; cb = c_below; ca = c_upto; }
            }
            while (ca - cb > uint32_t(kCap) && lo != hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                const uint32_t c = block_count_le<E, NW>(key, mid, sm, phase);
                if (c >= k) { hi = mid; ca = c; } else { lo = mid + 1; cb = c; }
            }
            const uint32_t need = k - cb;       // how many keys inside [lo,hi] are pruned (1..pop)
            const uint32_t pop = ca - cb;
            if (pop <= uint32_t(kCap)) {
                // exact rank among the <= kCap bracket keys; composite (key<<32 | col) is unique.
                // Compaction: ballot prefix inside the wave (no atomics), waves own LDS segments.
                const int wave = tid >> 6, lane = tid & 63;
                uint32_t base = 0;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const bool inb = key[i] >= lo && key[i] <= hi;
                    const unsigned long long b = __ballot(inb);
                    if (b) {
                        if (inb) {
                            const uint32_t pos = base + uint32_t(__popcll(b & ((1ull << lane) - 1ull)));
                            const uint32_t col = uint32_t((i / 8) * NT * 8 + (i % 8)) + tid8;
                            sm.seg[wave * kCap + pos] = (static_cast<unsigned long long>(key[i]) << 32) | col;
                        }
                        base += uint32_t(__popcll(b));
                    }
                }
                if constexpr (NW > 1) {
                    if (lane == 0) sm.segcnt[wave] = base;
                    lds_barrier();
                    if (tid < int(pop)) {          // gather the segments into one dense list
                        uint32_t t = uint32_t(tid);
                        int w = 0;
                        for (; w < NW - 1; ++w) {
                            const uint32_t c = sm.segcnt[w];
                            if (t < c) break;
                            t -= c;
                        }
                        sm.cand[tid] = sm.seg[w * kCap + t];
                    }
                }
                lds_barrier();
                if (tid < kWave) {
                    const unsigned long long *list = NW > 1 ? sm.cand : sm.seg;
                    const unsigned long long mine = tid < int(pop) ? list[tid] : ~0ull;
                    uint32_t rank = 0;
#pragma unroll
                    for (int j = 0; j < kCap; ++j) {
                        const unsigned long long other = list[j];            // wave-uniform address: broadcast
                        rank += (uint32_t(j) < pop && other < mine) ? 1u : 0u;
                    }
                    if (tid < int(pop) && rank == need - 1) sm.cut = mine;
                }
                lds_barrier();
                cut = sm.cut;
            } else {
                // lo == hi: more than kCap keys tie at the threshold value -> take the first `need`
                // of them in column order (stable sort semantics).
                uint32_t running = 0;
                if constexpr (NW > 1) lds_barrier();
#pragma unroll
                for (int s = 0; s < CH; ++s) {
                    uint32_t cnt = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) cnt += (key[s * 8 + j] == lo) ? 1u : 0u;
                    uint32_t incl = wave_incl_scan_u32(cnt);
                    uint32_t total = __shfl(incl, 63, kWave);
                    if constexpr (NW > 1) {
                        const int wave = tid >> 6;
                        if ((tid & 63) == 63) sm.scan[wave] = incl;
                        lds_barrier();
                        uint32_t before = 0, all = 0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            const uint32_t v = sm.scan[w];
                            before += (w < wave) ? v : 0u;
                            all += v;
                        }
                        incl += before;
                        total = all;
                        lds_barrier();
                    }
                    const uint32_t excl = incl - cnt;
                    if (running + excl < need && need <= running + incl) {
                        uint32_t target = need - running - excl;   // 1-based among this lane's equal keys
                        uint32_t colsel = 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (key[s * 8 + j] == lo) {
                                if (--target == 0) colsel = uint32_t(s * NT * 8 + j) + tid8;
                            }
                        }
                        sm.cut = (static_cast<unsigned long long>(lo) << 32) | colsel;
                    }
                    running += total;
                }
                lds_barrier();
                cut = sm.cut;
            }
            have_prev = true;
            prev_key = uint32_t(cut >> 32);
        }

        // ---- apply -----------------------------------------------------------------------
        uint8_t *mrow = mask + row * in_f;
        const uint32_t cut_key = uint32_t(cut >> 32), cut_col = uint32_t(cut);
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            if (!valid[s]) continue;
            const uint32_t c0 = uint32_t(s * NT * 8) + tid8;
            // column test `c0 + j <= cut_col` as `j <= rel`: keeps per-key column indices out of registers
            const int rel = (cut_col >= c0) ? int(cut_col - c0 > 8u ? 8u : cut_col - c0) : -1;
            uint32_t keepbits = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t kk = key[s * 8 + j];
                // (key, col) <= (cut_key, cut_col) in lexicographic order
                const bool pruned = (k > 0) && (kk < cut_key || (kk == cut_key && rel >= j));
                keepbits |= (pruned ? 0u : 1u) << j;
                if (pruned) raw[s].v[j] = typename T::raw(0);
            }
            store_mask_chunk<ALIGNED>(mrow, c0, in_f, keepbits);
            if (apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED>(wrow, c0, in_f, raw[s]);
        }

        // ---- row score sum (importance_score numerator) ------------------------------------
        if (row_sums) {
            double d = double(wave_sum_f32_dpp(fsum));
            if constexpr (NW > 1) {
                const int wave = tid >> 6;
                if ((tid & 63) == 0) sm.dsum[wave] = d;
                lds_barrier();
                d = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) d += sm.dsum[w];
            }
            if (tid == 0) row_sums[row] = d;
        }
        if constexpr (NW > 1) lds_barrier();   // smem reuse by the next row
    }
}

// ------------------------------------------------------------------------------------------
// SEL_MATRIX : global radix select (12 + 10 + 10 bits)
// ------------------------------------------------------------------------------------------
constexpr int kBits0 = 12, kBits1 = 10, kBits2 = 10;
constexpr int kBins0 = 1 << kBits0, kBins1 = 1 << kBits1, kBins2 = 1 << kBits2;
constexpr int kHistTotal = kBins0 + kBins1 + kBins2;

__device__ __forceinline__ int pass_bins(int p) { return p == 0 ? kBins0 : (p == 1 ? kBins1 : kBins2); }
__device__ __forceinline__ int pass_shift(int p) { return p == 0 ? 20 : (p == 1 ? 10 : 0); }
__device__ __forceinline__ int pass_off(int p) { return p == 0 ? 0 : (p == 1 ? kBins0 : kBins0 + kBins1); }

// Walk the finished histograms of passes [0, npass): returns the key prefix (bits above the next
// pass's digit) that contains rank `r`, and the rank left inside it.  Every workgroup recomputes
// this from the global histograms (a few KB from L2) instead of a separate tiny launch.
__device__ void resolve_prefix(const uint32_t *__restrict__ hist, int npass, uint64_t r, uint32_t *sh /*>=258 u32*/,
                               uint32_t &prefix, uint64_t &rank) {
    prefix = 0;
    rank = r;
    const int tid = threadIdx.x;
    for (int p = 0; p < npass; ++p) {
        const uint32_t *h = hist + pass_off(p);
        const int per = pass_bins(p) / 256;              // 16 or 4 bins per scanning thread
        if (tid < 256) {
            uint32_t a = 0;
            for (int i = 0; i < per; ++i) a += h[tid * per + i];
            sh[tid] = a;
        }
        __syncthreads();
        if (tid < 64) {                                   // wave 0: scan 64 groups of 4 partials
            const uint32_t a = sh[4 * tid] + sh[4 * tid + 1] + sh[4 * tid + 2] + sh[4 * tid + 3];
            const uint64_t incl = wave_incl_scan_u32(a);  // numel < 2^32
            const unsigned long long hit = __ballot(incl > rank);
            const int owner = hit ? __ffsll((long long)hit) - 1 : 63;
            if (tid == owner) {
                uint64_t cum = incl - a;
                int t = 4 * tid;
                for (; t < 4 * tid + 3; ++t) {
                    if (cum + sh[t] > rank) break;
                    cum += sh[t];
                }
                int b = t * per;
                for (; b < t * per + per - 1; ++b) {
                    const uint32_t c = h[b];
                    if (cum + c > rank) break;
                    cum += c;
                }
                sh[256] = uint32_t(b);
                sh[257] = uint32_t(rank - cum);
            }
        }
        __syncthreads();
        prefix |= sh[256] << pass_shift(p);
        rank = sh[257];
        __syncthreads();
    }
}

template <typename T, bool ALIGNED, int PASS>
__global__ __launch_bounds__(1024) void matrix_hist_kernel(const typename T::raw *__restrict__ W, int64_t out_f,
                                                           int64_t in_f, int64_t ldw, const float *__restrict__ sq,
                                                           uint64_t k_index, uint32_t *__restrict__ hist) {
    constexpr int BINS = PASS == 0 ? kBins0 : (PASS == 1 ? kBins1 : kBins2);
    constexpr int SHIFT = PASS == 0 ? 20 : (PASS == 1 ? 10 : 0);
    __shared__ uint32_t lh[BINS];
    __shared__ uint32_t sh[260];
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) lh[i] = 0;
    uint32_t prefix = 0;
    uint64_t rank = 0;
    resolve_prefix(hist, PASS, k_index, sh, prefix, rank);   // also the barrier after zeroing lh
    if constexpr (PASS == 0) __syncthreads();
    const uint32_t pmask = PASS == 0 ? 0u : (PASS == 1 ? 0xFFF00000u : 0xFFFFFC00u);

    const int64_t cpr = (in_f + 7) / 8;                     // chunks per row
    const int64_t total = out_f * cpr;
    for (int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; c < total; c += int64_t(gridDim.x) * blockDim.x) {
        const int64_t row = c / cpr, col0 = (c - row * cpr) * 8;
        Chunk8<T> raw = load_row_chunk<T, ALIGNED>(W + row * ldw, col0, in_f);
        float sqv[8];
        load_sq_chunk<ALIGNED>(sq, col0, in_f, sqv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (ALIGNED || col0 + j < in_f) {
                const uint32_t key = score_key(ieee_mul(fabsf(to_f32<T>(raw.v[j])), sqv[j]));
                if ((key & pmask) == prefix) atomicAdd(&lh[(key >> SHIFT) & (BINS - 1)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *gh = hist + (PASS == 0 ? 0 : (PASS == 1 ? kBins0 : kBins0 + kBins1));
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

template <typename T, bool ALIGNED>
__global__ __launch_bounds__(1024) void matrix_apply_kernel(typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f,
                                                            int64_t ldw, const float *__restrict__ sq, uint64_t k_index,
                                                            const uint32_t *__restrict__ hist, int apply_zero,
                                                            uint8_t *__restrict__ mask, double *__restrict__ block_sums) {
    __shared__ uint32_t sh[260];
    __shared__ double dsm[16];
    uint32_t thr = 0;
    uint64_t rank = 0;
    resolve_prefix(hist, 3, k_index, sh, thr, rank);
    // thr is the key of flat rank k_index; prune strictly below it.  A NaN threshold prunes nothing
    // (`score < nan` is False everywhere, wanda_pruner.py:683).
    const bool none = thr == 0xFFFFFFFFu;
    const int64_t cpr = (in_f + 7) / 8;
    const int64_t total = out_f * cpr;
    double dsum = 0.0;
    for (int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; c < total; c += int64_t(gridDim.x) * blockDim.x) {
        const int64_t row = c / cpr, col0 = (c - row * cpr) * 8;
        typename T::raw *wrow = W + row * ldw;
        Chunk8<T> raw = load_row_chunk<T, ALIGNED>(wrow, col0, in_f);
        float sqv[8];
        load_sq_chunk<ALIGNED>(sq, col0, in_f, sqv);
        uint32_t keepbits = 0;
        float fs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bool pruned = false;
            if (ALIGNED || col0 + j < in_f) {
                const float sc = ieee_mul(fabsf(to_f32<T>(raw.v[j])), sqv[j]);
                fs += sc;
                pruned = !none && score_key(sc) < thr;
            }
            keepbits |= (pruned ? 0u : 1u) << j;
            if (pruned) raw.v[j] = typename T::raw(0);
        }
        dsum += double(fs);
        store_mask_chunk<ALIGNED>(mask + row * in_f, col0, in_f, keepbits);
        if (apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED>(wrow, col0, in_f, raw);
    }
    if (block_sums) {
        dsum = wave_sum_f64(dsum);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) dsm[wave] = dsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0;
            for (int w = 0; w < int(blockDim.x >> 6); ++w) a += dsm[w];
            block_sums[blockIdx.x] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------
// SEL_NM
// ------------------------------------------------------------------------------------------
template <typename T, bool ALIGNED, int M>
__global__ __launch_bounds__(256) void nm_kernel(typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f, int64_t ldw,
                                                 const float *__restrict__ sq, int n, int apply_zero,
                                                 uint8_t *__restrict__ mask, double *__restrict__ block_sums) {
    __shared__ double dsm[4];
    const int64_t cpr = (in_f + 7) / 8;
    const int64_t total = out_f * cpr;
    double dsum = 0.0;
    for (int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; c < total; c += int64_t(gridDim.x) * blockDim.x) {
        const int64_t row = c / cpr, col0 = (c - row * cpr) * 8;
        typename T::raw *wrow = W + row * ldw;
        Chunk8<T> raw = load_row_chunk<T, ALIGNED>(wrow, col0, in_f);
        float sqv[8];
        load_sq_chunk<ALIGNED>(sq, col0, in_f, sqv);
        uint32_t key[8];
        float fs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (ALIGNED || col0 + j < in_f) {
                const float sc = ieee_mul(fabsf(to_f32<T>(raw.v[j])), sqv[j]);
                fs += sc;
                key[j] = score_key(sc);
            } else {
                key[j] = 0xFFFFFFFFu;
            }
        }
        dsum += double(fs);
        uint32_t keepbits = 0;
#pragma unroll
        for (int g = 0; g < 8 / M; ++g) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                // stable rank of column i inside its group: smaller key first, then lower index
                int rank = 0;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const uint32_t kj = key[g * M + j], ki = key[g * M + i];
                    rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
                }
                const bool pruned = rank < n && (ALIGNED || col0 + g * M + i < in_f);
                keepbits |= (pruned ? 0u : 1u) << (g * M + i);
                if (pruned) raw.v[g * M + i] = typename T::raw(0);
            }
        }
        store_mask_chunk<ALIGNED>(mask + row * in_f, col0, in_f, keepbits);
        if (apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED>(wrow, col0, in_f, raw);
    }
    if (block_sums) {
        dsum = wave_sum_f64(dsum);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) dsm[wave] = dsum;
        __syncthreads();
        if (threadIdx.x == 0) block_sums[blockIdx.x] = dsm[0] + dsm[1] + dsm[2] + dsm[3];
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr int kMatrixGrid = 512;     // persistent 1024-thread workgroups (2 per CU)
constexpr int kNmGrid = 2048;        // 256-thread workgroups (8 per CU)

struct WsLayout {
    size_t hist_off, total;
};
static WsLayout ws_layout(int mode, int64_t out_f, int64_t in_f) {
    WsLayout l{};
    (void)out_f; (void)in_f;
    if (mode == VLMC_SEL_MATRIX) l.total = round_up(size_t(kHistTotal) * 4, 256);   // radix histograms
    return l;
}
static int64_t n_partials(int mode, int64_t out_f) {
    return mode == VLMC_SEL_ROW ? out_f : (mode == VLMC_SEL_MATRIX ? kMatrixGrid : kNmGrid);
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <typename T, int CH, int NW, bool ALIGNED>
static void launch_rows(void *W, int64_t out_f, int64_t in_f, int64_t ldw, const float *sqrt_scaler, uint32_t k,
                        int apply_zero, uint8_t *mask, double *row_sums, hipStream_t st) {
    // persistent grid: ~16 waves per CU, every workgroup walks the same number of rows (+-1)
    const int waves_per_cu = env_int("VLMC_SELECT_WAVES_PER_CU", 16);
    int64_t max_grid = int64_t(256) * waves_per_cu / NW;
    if (max_grid < 1) max_grid = 1;
    const int64_t rows_per_wg = (out_f + max_grid - 1) / max_grid;
    const int64_t grid = (out_f + rows_per_wg - 1) / rows_per_wg;
    hipLaunchKernelGGL((select_rows_kernel<T, CH, NW, ALIGNED>), dim3(unsigned(grid)), dim3(64 * NW), 0, st,
                       static_cast<typename T::raw *>(W), out_f, in_f, ldw, sqrt_scaler, k, apply_zero, mask, row_sums);
}

template <typename T>
static int dispatch_rows(void *W, int64_t out_f, int64_t in_f, int64_t ldw, const float *sqrt_scaler, uint32_t k, int apply_zero,
                         uint8_t *mask, double *row_sums, bool aligned, hipStream_t st) {
    const int64_t nchunks = (in_f + 7) / 8;
#define VLMC_ROWS(CH, NW, AL) launch_rows<T, CH, NW, AL>(W, out_f, in_f, ldw, sqrt_scaler, k, apply_zero, mask, row_sums, st)
    if (!aligned) {
        if (nchunks <= 256) VLMC_ROWS(4, 1, false);
        else if (nchunks <= 2048) VLMC_ROWS(4, 8, false);
        else { set_error("vlmc_wanda_select: in_features %lld too large (max 16384)", (long long)in_f); return VLMC_EINVAL; }
        return VLMC_OK;
    }
    int nw = 1;
    while (nw < 8 && nchunks > int64_t(64) * nw * 4) nw *= 2;
    if (nchunks > int64_t(64) * nw * 4) {
        set_error("vlmc_wanda_select: in_features %lld too large (max 16384)", (long long)in_f);
        return VLMC_EINVAL;
    }
    const int ch = int((nchunks + 64 * nw - 1) / (64 * nw));
#define VLMC_ROWS_NW(NW)                          \
    switch (ch) {                                 \
        case 1: VLMC_ROWS(1, NW, true); break;    \
        case 2: VLMC_ROWS(2, NW, true); break;    \
        case 3: VLMC_ROWS(3, NW, true); break;    \
        default: VLMC_ROWS(4, NW, true); break;   \
    }
    switch (nw) {
        case 1: VLMC_ROWS_NW(1); break;
        case 2: VLMC_ROWS_NW(2); break;
        case 4: VLMC_ROWS_NW(4); break;
        default: VLMC_ROWS_NW(8); break;
    }
#undef VLMC_ROWS_NW
#undef VLMC_ROWS
    return VLMC_OK;
}

template <typename T, bool AL>
static void launch_matrix(void *W, int64_t out_f, int64_t in_f, int64_t ldw, const float *sq, uint64_t k_index,
                          int apply_zero, uint8_t *mask, double *parts, uint32_t *hist, hipStream_t st) {
    using raw = typename T::raw;
    raw *w = static_cast<raw *>(W);
    (void)hipMemsetAsync(hist, 0, size_t(kHistTotal) * 4, st);
    hipLaunchKernelGGL((matrix_hist_kernel<T, AL, 0>), dim3(kMatrixGrid), dim3(1024), 0, st, w, out_f, in_f, ldw, sq, k_index, hist);
    hipLaunchKernelGGL((matrix_hist_kernel<T, AL, 1>), dim3(kMatrixGrid), dim3(1024), 0, st, w, out_f, in_f, ldw, sq, k_index, hist);
    hipLaunchKernelGGL((matrix_hist_kernel<T, AL, 2>), dim3(kMatrixGrid), dim3(1024), 0, st, w, out_f, in_f, ldw, sq, k_index, hist);
    hipLaunchKernelGGL((matrix_apply_kernel<T, AL>), dim3(kMatrixGrid), dim3(1024), 0, st, w, out_f, in_f, ldw, sq, k_index,
                       hist, apply_zero, mask, parts);
}

template <typename T, bool AL>
static int launch_nm(void *W, int64_t out_f, int64_t in_f, int64_t ldw, const float *sq, int n, int m, int apply_zero,
                     uint8_t *mask, double *parts, hipStream_t st) {
    using raw = typename T::raw;
    raw *w = static_cast<raw *>(W);
#define VLMC_NM(M) hipLaunchKernelGGL((nm_kernel<T, AL, M>), dim3(kNmGrid), dim3(256), 0, st, w, out_f, in_f, ldw, sq, n, apply_zero, mask, parts)
    switch (m) {
        case 2: VLMC_NM(2); break;
        case 4: VLMC_NM(4); break;
        case 8: VLMC_NM(8); break;
        default: set_error("vlmc_wanda_select: n:m with m=%d unsupported (m must be 2, 4 or 8)", m); return VLMC_EINVAL;
    }
#undef VLMC_NM
    return VLMC_OK;
}

template <typename T>
static int select_typed(void *W, int64_t out_f, int64_t in_f, int64_t ldw, const float *sqrt_scaler, int mode, int64_t k,
                        int n, int m, int apply_zero, uint8_t *mask, double *parts, char *ws, hipStream_t st) {
    const WsLayout l = ws_layout(mode, out_f, in_f);
    const bool aligned = in_f % 8 == 0 && ldw % 8 == 0 && aligned16(W) && aligned16(sqrt_scaler) &&
                         (reinterpret_cast<uintptr_t>(mask) % 8) == 0;
    int rc = VLMC_OK;
    if (mode == VLMC_SEL_ROW) {
        rc = dispatch_rows<T>(W, out_f, in_f, ldw, sqrt_scaler, uint32_t(k), apply_zero, mask, parts, aligned, st);
    } else {
        const float *sq = sqrt_scaler;
        uint32_t *hist = reinterpret_cast<uint32_t *>(ws + l.hist_off);
        if (mode == VLMC_SEL_MATRIX) {
            if (aligned) launch_matrix<T, true>(W, out_f, in_f, ldw, sq, uint64_t(k), apply_zero, mask, parts, hist, st);
            else launch_matrix<T, false>(W, out_f, in_f, ldw, sq, uint64_t(k), apply_zero, mask, parts, hist, st);
        } else {
            rc = aligned ? launch_nm<T, true>(W, out_f, in_f, ldw, sq, n, m, apply_zero, mask, parts, st)
                         : launch_nm<T, false>(W, out_f, in_f, ldw, sq, n, m, apply_zero, mask, parts, st);
        }
    }
    if (rc != VLMC_OK) return rc;
    VLMC_HIP_CHECK_LAUNCH("vlmc_wanda_select");
    return VLMC_OK;
}

}  // namespace vlmc

using namespace vlmc;

extern "C" size_t vlmc_wanda_select_workspace(int mode, int64_t out_features, int64_t in_features) {
    if (out_features <= 0 || in_features <= 0 || mode < 0 || mode > 2) return 0;
    return ws_layout(mode, out_features, in_features).total;
}

extern "C" int64_t vlmc_wanda_select_partials(int mode, int64_t out_features, int64_t in_features) {
    if (out_features <= 0 || in_features <= 0 || mode < 0 || mode > 2) return 0;
    return n_partials(mode, out_features);
}

extern "C" int vlmc_wanda_select(void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw,
                                 const float *sqrt_scaler, int mode, int64_t k, int n, int m, int apply_zero,
                                 uint8_t *mask, double *score_partials, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    VLMC_REQUIRE(W && sqrt_scaler && mask, "vlmc_wanda_select: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && ldw >= in_features,
                 "vlmc_wanda_select: bad shape out=%lld in=%lld ldw=%lld", (long long)out_features, (long long)in_features,
                 (long long)ldw);
    VLMC_REQUIRE(out_features * in_features < (int64_t(1) << 32), "vlmc_wanda_select: more than 2^32 weights");
    VLMC_REQUIRE(mode >= 0 && mode <= 2, "vlmc_wanda_select: unknown mode %d", mode);
    if (mode == VLMC_SEL_ROW) {
        VLMC_REQUIRE(k >= 0 && k <= in_features, "vlmc_wanda_select: row k=%lld outside [0,%lld]", (long long)k,
                     (long long)in_features);
    } else if (mode == VLMC_SEL_MATRIX) {
        VLMC_REQUIRE(k >= 0 && k < out_features * in_features, "vlmc_wanda_select: matrix k=%lld outside [0,%lld)",
                     (long long)k, (long long)(out_features * in_features));
    } else {
        VLMC_REQUIRE(m > 0 && n >= 0 && n <= m && in_features % m == 0,
                     "vlmc_wanda_select: bad n:m = %d:%d for in_features %lld", n, m, (long long)in_features);
    }
    const size_t need = vlmc_wanda_select_workspace(mode, out_features, in_features);
    if (need) {
        VLMC_REQUIRE(workspace, "vlmc_wanda_select: null workspace");
        VLMC_REQUIRE((reinterpret_cast<uintptr_t>(workspace) % 256) == 0, "vlmc_wanda_select: workspace not 256-B aligned");
        if (workspace_bytes < need) {
            set_error("vlmc_wanda_select: workspace %zu B < required %zu B", workspace_bytes, need);
            return VLMC_EWORKSPACE;
        }
    }
    hipStream_t st = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    switch (dtype) {
        case VLMC_F32: return select_typed<f32_t>(W, out_features, in_features, ldw, sqrt_scaler, mode, k, n, m, apply_zero, mask, score_partials, ws, st);
        case VLMC_F16: return select_typed<f16_t>(W, out_features, in_features, ldw, sqrt_scaler, mode, k, n, m, apply_zero, mask, score_partials, ws, st);
        case VLMC_BF16: return select_typed<bf16_t>(W, out_features, in_features, ldw, sqrt_scaler, mode, k, n, m, apply_zero, mask, score_partials, ws, st);
    }
    set_error("vlmc_wanda_select: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}
