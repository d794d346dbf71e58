// K2-K7: fused Wanda score + mask selection + apply for one linear
// (replaces /root/reference/lavis/compression/pruners/wanda_pruner.py:318-341 and :666-687).
//
// HBM-bound: per weight element the algorithm must read W (2 B), write the bool mask (1 B) and
// write the zeroed W back (2 B) => 5 B/weight (3 B under lora_model=True).  The fp32 score
// |W|*sqrt(s) is never materialised; it lives in registers as an order-preserving u32 key.
//
//  SEL_ROW    one workgroup (1..8 waves) per row, grid = rows; the row's keys stay in VGPRs
//             (lanes own 16-byte column chunks, so loads/stores are coalesced).  The k-th smallest
//             key is found by radix-64 refinement of a bracket with 64 LDS counters (6 key bits
//             per sweep), seeded by a 32-key sample of the row; ties are broken by column index
//             exactly like the reference's stable sort.  See the kernel's header comment.
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
        // spread bit j to byte j: (nibble * 0x204081) & 0x01010101 puts bits 0..3 into bytes 0..3
        uint2 m;
        m.x = ((keepbits & 0xFu) * 0x00204081u) & 0x01010101u;
        m.y = (((keepbits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
        *reinterpret_cast<uint2 *>(mrow + col0) = m;
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
// SEL_ROW: one workgroup of NW waves per row, grid = rows (the hardware dispatcher overlaps
// rows that are loading, searching and storing on every SIMD).
//
//   1. keys in registers (8*CH per lane, 64*NW lanes; lanes own 16-byte column chunks).
//   2. threshold search = radix-64 refinement of a bracket [lo, lo + 64*2^shift): every key
//      inside the bracket bumps one of 64 LDS counters (bin = (key-lo) >> shift; keys outside
//      go to a private per-thread sink slot, so there is no bank conflict and no divergence),
//      a DPP prefix scan over 64 lanes finds the bin holding rank k, and that bin becomes the
//      next bracket -- 6 key bits per sweep instead of 1 per bisection step.
//      First bracket: two order statistics of a 32-key sample of the row (all-pairs ranks via
//      readlane) bound the threshold to ~45 % of the keys, which spreads the first sweep's
//      atomics over the bins; the sweep itself verifies the guess by exact counting and falls
//      back to the whole key range if it was wrong.
//      The refinement stops when every key left in the bracket is pruned (need == pop) or the
//      bracket is a single value (ties, resolved in column order like the reference's stable sort).
//   3. apply: one compare per key -> bool mask bytes + zeroed weights.
// All decisions rest on exact counts: the sample only affects speed, never the result.
// ------------------------------------------------------------------------------------------
#ifdef VLMC_STAMPS
// diagnostic build only (never shipped): per-phase cycle totals, summed over waves into row_sums[0..7]
#define VLMC_STAMP(i)                                                                   \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        unsigned long long t_;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                              \
        stamp_acc[i] += t_ - stamp_last;                                                \
        stamp_last = t_;                                                                \
    } while (0)
#else
#define VLMC_STAMP(i) do {} while (0)
#endif

// inclusive prefix sum across the 64 lanes with DPP row shifts + row broadcasts
__device__ __forceinline__ uint32_t wave_incl_scan_u32_dpp(uint32_t v) {
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, false));   // row_shr:1
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, false));   // row_shr:2
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, false));   // row_shr:4
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, false));   // row_shr:8
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1,3
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
    v = max(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0xB1, 0xF, 0xF, true)));
    v = max(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x4E, 0xF, 0xF, true)));
    v = max(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x141, 0xF, 0xF, true)));
    v = max(v, uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x140, 0xF, 0xF, true)));
    return max(max(uint32_t(__builtin_amdgcn_readlane(int(v), 0)), uint32_t(__builtin_amdgcn_readlane(int(v), 16))),
               max(uint32_t(__builtin_amdgcn_readlane(int(v), 32)), uint32_t(__builtin_amdgcn_readlane(int(v), 48))));
}
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) { return ~wave_max_u32_dpp(~v); }

constexpr int kBins = 64;
constexpr int kSample = 32;

template <int NW> struct RowSmem {
    uint32_t hist[kBins + 64 * NW];   // [0,64): bins; [64, 64+NT): per-thread sinks
    uint32_t sample[kSample];
    uint32_t below[NW];               // per-wave count(key < lo) of the guessed sweep
    uint32_t scan[NW];
    float fsum[NW];
    uint32_t cut_col;
};

// workgroup-wide sync of LDS traffic; a single-wave workgroup only needs program order
template <int NW> __device__ __forceinline__ void row_sync() {
    if constexpr (NW > 1) {
        lds_barrier();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, int CH, int NW, bool ALIGNED>
__global__ __launch_bounds__(64 * NW, (NW * CH >= 4 && CH >= 3) ? 5 : 8)
void select_rows_kernel(typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f, int64_t ldw,
                        const float *__restrict__ sqrt_scaler, uint32_t k, int apply_zero, uint8_t *__restrict__ mask,
                        double *__restrict__ row_sums, uint32_t sample_margin, uint32_t frac_q16) {
    constexpr int NT = 64 * NW;
    constexpr int E = CH * 8;
    __shared__ RowSmem<NW> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nchunks = (in_f + 7) / 8;
    const int64_t row = blockIdx.x;
    typename T::raw *wrow = W + row * ldw;
    const uint32_t sink = uint32_t(kBins + tid);
    const uint32_t tid8 = uint32_t(tid) * 8u;
#ifdef VLMC_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

    // ---- 1. load the row, build the keys ------------------------------------------------------
    bool valid[CH];
    Chunk8<T> raw[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        valid[s] = int64_t(s) * NT + tid < nchunks;
        if (valid[s]) raw[s] = load_row_chunk<T, ALIGNED>(wrow, (int64_t(s) * NT + tid) * 8, in_f);
    }
    VLMC_STAMP(0);
    uint32_t key[E];
    float fsum = 0.f;
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        const int64_t col0 = (int64_t(s) * NT + tid) * 8;
        float sq[8];
        if (valid[s]) load_sq_chunk<ALIGNED>(sqrt_scaler, col0, in_f, sq);     // <= 64 KB table, L2-resident
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = valid[s] && (ALIGNED || col0 + j < in_f);
            if (live) {
                const float sc = ieee_mul(fabsf(to_f32<T>(raw[s].v[j])), sq[j]);
                // score >= +0 or NaN; every NaN -> one key above +inf (torch.sort puts NaN last)
                const uint32_t b = __float_as_uint(sc);
                key[s * 8 + j] = b < 0x7F800001u ? b : 0x7F800001u;
                fsum += sc;
            } else {
                key[s * 8 + j] = 0xFFFFFFFFu;   // padding sorts after every real column
            }
        }
    }
    VLMC_STAMP(1);

    uint32_t cut_key = 0, cut_col = 0;
    if (k > 0) {
        // ---- 2a. bracket guess from a 32-key sample ------------------------------------------------
        uint32_t lo = 0, cb = 0, shift = 26;          // default: whole key range, 64 * 2^26 = 2^32
        bool guessed = false;
        {
            // sample = first key of slot (lane mod CH) of the first 32 threads: spread over the columns
            uint32_t v = key[0];
#pragma unroll
            for (int s = 1; s < CH; ++s) v = (lane % CH == s) ? key[s * 8] : v;
            if constexpr (NW > 1) {
                if (tid < kSample) sm.sample[tid] = v;
                lds_barrier();
                v = sm.sample[lane & (kSample - 1)];
            }
            const bool in_sample = lane < kSample && v != 0xFFFFFFFFu;
            const uint32_t nsample = uint32_t(__popcll(__ballot(in_sample)));
            if (!in_sample) v = 0xFFFFFFFFu;
            uint32_t rank = 0, rank2 = 0;
#pragma unroll
            for (int j = 0; j < kSample; j += 2) {
                rank += (uint32_t(__builtin_amdgcn_readlane(int(v), j)) < v) ? 1u : 0u;
                rank2 += (uint32_t(__builtin_amdgcn_readlane(int(v), j + 1)) < v) ? 1u : 0u;
            }
            rank += rank2;
            const uint32_t rs = (frac_q16 * nsample) >> 16;                            // sample rank of the target
            const uint32_t r_lo = rs > sample_margin ? rs - sample_margin : 0u;
            const uint32_t r_hi = rs + sample_margin;
            const uint32_t glo = r_lo == 0 ? 0u : wave_max_u32_dpp((rank <= r_lo && in_sample) ? v : 0u);
            uint32_t ghi = wave_min_u32_dpp((rank >= r_hi && in_sample) ? v : 0xFFFFFFFFu);
            if (ghi > 0x7F800001u) ghi = 0x7F800001u;                                 // largest real key
            if (ghi > glo) {
                const uint32_t w = (ghi - glo) >> 6;                                  // 64 bins must cover [glo, ghi]
                shift = w ? 32u - uint32_t(__builtin_clz(w)) : 0u;
                lo = glo;
                if (shift < 26u) {
                    const uint64_t end = uint64_t(lo) + (uint64_t(64) << shift);
                    if (end > 0x100000000ull) lo = uint32_t(0x100000000ull - (uint64_t(64) << shift));
                    guessed = true;
                } else {
                    lo = 0; shift = 26;
                }
            }
        }
        VLMC_STAMP(2);
        // ---- 2b. radix-64 bracket refinement ------------------------------------------------------
        // invariant: cb = count(key < lo) < k <= count(key < lo + 64*2^shift)
        uint32_t need = 0, pop = 0;
        for (;;) {
            if (tid < kBins) sm.hist[tid] = 0;
            uint32_t below = 0, below2 = 0;
            if (guessed) {
#pragma unroll
                for (int i = 0; i < E; i += 2) {
                    below += (key[i] < lo) ? 1u : 0u;
                    below2 += (key[i + 1] < lo) ? 1u : 0u;
                }
                if constexpr (NW > 1) {
                    const uint32_t wsum = wave_sum_u32_dpp(below + below2);
                    if (lane == 0) sm.below[wave] = wsum;
                }
            }
            if constexpr (NW > 1) lds_barrier();          // bins cleared before anybody adds
            // bin = (key - lo) >> shift; anything outside the bracket (including wrapped key < lo)
            // is >= 64 and lands in this thread's private sink slot
#pragma unroll
            for (int i = 0; i < E; i += 4) {
                uint32_t a0 = (key[i] - lo) >> shift, a1 = (key[i + 1] - lo) >> shift;
                uint32_t a2 = (key[i + 2] - lo) >> shift, a3 = (key[i + 3] - lo) >> shift;
                a0 = min(a0, sink); a1 = min(a1, sink); a2 = min(a2, sink); a3 = min(a3, sink);
                atomicAdd(&sm.hist[a0], 1u); atomicAdd(&sm.hist[a1], 1u);
                atomicAdd(&sm.hist[a2], 1u); atomicAdd(&sm.hist[a3], 1u);
            }
            row_sync<NW>();
            if (guessed) {
                if constexpr (NW > 1) {
                    cb = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) cb += sm.below[w];
                } else {
                    cb = wave_sum_u32_dpp(below + below2);
                }
            }
            const uint32_t h = sm.hist[lane];                        // every wave scans the 64 bins itself
            const uint32_t incl = wave_incl_scan_u32_dpp(h);
            const unsigned long long hit = __ballot(cb + incl >= k);
            if constexpr (NW > 1) lds_barrier();                     // bins read before the next sweep clears them
            if (guessed) {
                guessed = false;
                if (cb >= k || hit == 0) {                          // wrong guess: start over on the full range
                    lo = 0; cb = 0; shift = 26;
                    continue;
                }
            }
            const int b = __builtin_ctzll(hit);                     // bin that holds rank k
            const uint32_t incl_b = uint32_t(__builtin_amdgcn_readlane(int(incl), b));
            pop = uint32_t(__builtin_amdgcn_readlane(int(h), b));
            cb += incl_b - pop;
            need = k - cb;                                           // 1..pop
            lo += uint32_t(b) << shift;
            if (need == pop || shift == 0) break;
            shift = shift >= 6 ? shift - 6 : 0;
        }
        VLMC_STAMP(3);
        // bracket is now [lo, lo + 2^shift): all `pop` keys inside it are pruned, or shift == 0
        // and `need` of the `pop` keys equal to lo are pruned (first by column).
        cut_key = lo + ((1u << shift) - 1u);
        if (need == pop) {
            cut_col = 0xFFFFFFFFu;
        } else {
            // columns grow with (slot, thread, j): ordered count of the keys equal to lo
            uint32_t running = 0;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                uint32_t cnt = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) cnt += (key[s * 8 + j] == lo) ? 1u : 0u;
                uint32_t incl = wave_incl_scan_u32_dpp(cnt);
                uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
                if constexpr (NW > 1) {
                    if (lane == 63) sm.scan[wave] = incl;
                    lds_barrier();
                    uint32_t before = 0, all = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const uint32_t t = sm.scan[w];
                        before += (w < wave) ? t : 0u;
                        all += t;
                    }
                    incl += before;
                    total = all;
                    lds_barrier();
                }
                const uint32_t excl = incl - cnt;
                if (running + excl < need && need <= running + incl) {
                    uint32_t target = need - running - excl;
                    uint32_t colsel = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (key[s * 8 + j] == lo && --target == 0) colsel = uint32_t(s * NT * 8 + j) + tid8;
                    sm.cut_col = colsel;
                }
                running += total;
            }
            row_sync<NW>();
            cut_col = sm.cut_col;
        }
    }
    VLMC_STAMP(4);

    // ---- 3. apply -------------------------------------------------------------------------------
    uint8_t *mrow = mask + row * in_f;
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        if (!valid[s]) continue;
        const uint32_t c0 = uint32_t(s * NT * 8) + tid8;
        uint32_t keepbits = 0;
        if (k == 0) {
            keepbits = 0xFFu;
        } else if (cut_key < 0xFFFFFFFFu && !(c0 <= cut_col && cut_col < c0 + 7u)) {
            // whole chunk on one side of the tie column: one compare per key
            const uint32_t t = cut_key + ((c0 + 7u <= cut_col) ? 1u : 0u);    // prune key < t
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool keep = key[s * 8 + j] >= t;
                keepbits |= (keep ? 1u : 0u) << j;
                raw[s].v[j] = keep ? raw[s].v[j] : typename T::raw(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t kk = key[s * 8 + j];
                const bool keep = !(kk < cut_key || (kk == cut_key && c0 + uint32_t(j) <= cut_col));
                keepbits |= (keep ? 1u : 0u) << j;
                raw[s].v[j] = keep ? raw[s].v[j] : typename T::raw(0);
            }
        }
        store_mask_chunk<ALIGNED>(mrow, c0, in_f, keepbits);
        if (apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED>(wrow, c0, in_f, raw[s]);
    }
    VLMC_STAMP(5);
#ifndef VLMC_STAMPS
    if (row_sums) {
        const float tot = wave_sum_f32_dpp(fsum);
        if constexpr (NW > 1) {
            if (lane == 0) sm.fsum[wave] = tot;
            lds_barrier();
            if (tid == 0) {
                double d = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) d += double(sm.fsum[w]);
                row_sums[row] = d;
            }
        } else {
            if (lane == 0) row_sums[row] = double(tot);
        }
    }
#else
    if (tid == 0 && row_sums) {
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(row_sums) + i, stamp_acc[i]);
        atomicAdd(reinterpret_cast<unsigned long long *>(row_sums) + 8, 1ull);
    }
#endif
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
    const uint32_t margin = uint32_t(env_int("VLMC_SELECT_SAMPLE_MARGIN", 9));   // ~ +-3 sigma of a 32-sample rank
    const uint32_t frac_q16 = uint32_t((uint64_t(k) << 16) / uint64_t(in_f));
    hipLaunchKernelGGL((select_rows_kernel<T, CH, NW, ALIGNED>), dim3(unsigned(out_f)), dim3(64 * NW), 0, st,
                       static_cast<typename T::raw *>(W), out_f, in_f, ldw, sqrt_scaler, k, apply_zero, mask, row_sums,
                       margin, frac_q16);
}

// (NW waves) x (CH chunks of 8 columns per lane) must cover the row: 64*NW*CH >= in/8.
// Few rows -> more waves per row (parallelism); many rows -> one wave per row (no barriers).
template <typename T>
static int dispatch_rows(void *W, int64_t out_f, int64_t in_f, int64_t ldw, const float *sqrt_scaler, uint32_t k, int apply_zero,
                         uint8_t *mask, double *row_sums, bool aligned, hipStream_t st) {
    const int64_t nchunks = (in_f + 7) / 8;
    if (nchunks > 2048) {
        set_error("vlmc_wanda_select: in_features %lld too large (max 16384)", (long long)in_f);
        return VLMC_EINVAL;
    }
#define VLMC_ROWS(CH, NW, AL) launch_rows<T, CH, NW, AL>(W, out_f, in_f, ldw, sqrt_scaler, k, apply_zero, mask, row_sums, st)
    if (!aligned) {
        if (nchunks <= 256) VLMC_ROWS(4, 1, false);
        else VLMC_ROWS(4, 8, false);
        return VLMC_OK;
    }
    int nw = 1;
    while (nw < 8 && nchunks > int64_t(64) * nw * 4) nw *= 2;               // CH <= 4
    const int64_t want_waves = env_int("VLMC_SELECT_WANT_WAVES", 4096);      // ~4 waves per SIMD
    while (nw < 8 && out_f * nw < want_waves && nchunks > int64_t(64) * nw) nw *= 2;
    if (const int f = env_int("VLMC_SELECT_NW", 0)) {
        if ((f == 1 || f == 2 || f == 4 || f == 8) && nchunks <= int64_t(64) * f * 4) nw = f;
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
