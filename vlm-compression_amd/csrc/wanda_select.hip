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
//             exactly like the reference's stable sort.  See the kernel's header comment.  Rows of two widths
//             (a T5 block: 2048 and 5120 columns) share one launch of 4-wave workgroups (select_rows_mixed_kernel).
//  SEL_MATRIX one kernel of co-resident workgroups that keep W in registers (matrix_fused_kernel): sample ->
//             bracket, 2048-bin histogram of the bracket, grid barrier, bin of rank k (refined from the registers if
//             crowded), candidate keys exchanged through per-workgroup slots, grid barrier, exact threshold, apply.
//             The four-launch form (sample / count / apply / resolve) remains for in_features > 8192 and as the
//             cross-check.  See the section comments below.
//  SEL_NM     elementwise: each lane ranks the columns of its m-groups in registers.
// Every kernel takes a table of up to 12 linears ("jobs") by value, so all linears of a transformer
// block that share a launch shape go to the GPU in ONE launch.
#include <cstdlib>

#include "common.hpp"
#include "topk_order.hpp"

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
template <typename T, bool ALIGNED, bool NT = false>
__device__ __forceinline__ Chunk8<T> load_row_chunk(const typename T::raw *row, int64_t col0, int64_t in_f) {
    if constexpr (ALIGNED) {
        return load_chunk8<T, NT>(row + col0);
    } else {
        Chunk8<T> c;
#pragma unroll
        for (int j = 0; j < 8; ++j) c.v[j] = (col0 + j < in_f) ? row[col0 + j] : typename T::raw(0);
        return c;
    }
}
template <typename T, bool ALIGNED, bool NT = false>
__device__ __forceinline__ void store_row_chunk(typename T::raw *row, int64_t col0, int64_t in_f, const Chunk8<T> &c) {
    if constexpr (ALIGNED) {
        store_chunk8<T, NT>(row + col0, c);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (col0 + j < in_f) row[col0 + j] = c.v[j];
    }
}
template <bool ALIGNED, bool NT = false>
__device__ __forceinline__ void store_mask_chunk(uint8_t *mrow, int64_t col0, int64_t in_f, uint32_t keepbits) {
    if constexpr (ALIGNED) {
        // spread bit j to byte j: (nibble * 0x204081) & 0x01010101 puts bits 0..3 into bytes 0..3
        u32x2_t m;
        m.x = ((keepbits & 0xFu) * 0x00204081u) & 0x01010101u;
        m.y = (((keepbits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
        u32x2_t *q = reinterpret_cast<u32x2_t *>(mrow + col0);
        if constexpr (NT) __builtin_nontemporal_store(m, q); else *q = m;
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
// job table (kernel argument, passed by value)
// ------------------------------------------------------------------------------------------
constexpr int kMaxSelJobs = 12;
constexpr int kMatrixParts = 512;   // score partial sums per SEL_MATRIX job (= max workgroups per job)
constexpr int kNmParts = 2048;      // ... per SEL_NM job
struct SelJob {
    void *W;
    const float *sq;          // sqrt(scaler_row) [in]
    uint8_t *mask;            // [out, in]
    double *parts;            // score partial sums (may be null)
    uint32_t *ws;             // SEL_MATRIX workspace
    uint32_t out_f, in_f, ldw;
    uint32_t k;               // SEL_ROW: columns pruned per row; SEL_MATRIX: flat rank of the threshold
    uint32_t unit_end;        // exclusive prefix end of this job's grid units (rows or workgroups)
    uint32_t nwg;             // workgroups of this job in the elementwise passes
};
struct SelBatch {
    SelJob job[kMaxSelJobs];
    int32_t n, apply_zero;
    uint32_t p0, p1;          // SEL_ROW: sample margin, k/in as Q16; SEL_NM: n; SEL_MATRIX: force-slow flag
};
__device__ __forceinline__ int find_job(const SelBatch &b, uint32_t unit, uint32_t &local) {
    int j = 0;
    while (j + 1 < b.n && unit >= b.job[j].unit_end) ++j;
    local = unit - (j ? b.job[j - 1].unit_end : 0u);
    return j;
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

// One row by the NW waves whose lanes are numbered `tid` = 0 .. 64*NW-1 and who own `sm`.
template <typename T, int CH, int NW, bool ALIGNED>
__device__ __forceinline__ void select_row_body(const SelBatch &b, const SelJob &jb, const uint32_t row32, RowSmem<NW> &sm,
                                                const int tid) {
    constexpr int NT = 64 * NW;
    constexpr int E = CH * 8;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t in_f = jb.in_f, row = row32;
    const float *__restrict__ sqrt_scaler = jb.sq;
    uint8_t *__restrict__ mask = jb.mask;
    double *__restrict__ row_sums = jb.parts;
    const uint32_t k = jb.k, sample_margin = b.p0, frac_q16 = jb.nwg /* k/in as Q16, set by the host */;
    const int apply_zero = b.apply_zero;
    const int64_t nchunks = (in_f + 7) / 8;
    typename T::raw *wrow = static_cast<typename T::raw *>(jb.W) + row * int64_t(jb.ldw);
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
        if (valid[s]) raw[s] = load_row_chunk<T, ALIGNED, true>(wrow, (int64_t(s) * NT + tid) * 8, in_f);
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
        store_mask_chunk<ALIGNED, true>(mrow, c0, in_f, keepbits);
        if (apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED, true>(wrow, c0, in_f, raw[s]);
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

template <typename T, int CH, int NW, bool ALIGNED>
__global__ __launch_bounds__(64 * NW, (NW * CH >= 4 && CH >= 3) ? 5 : 8)
void select_rows_kernel(const SelBatch b) {
    __shared__ RowSmem<NW> sm;
    uint32_t row32;
    const SelJob &jb = b.job[find_job(b, blockIdx.x, row32)];
    select_row_body<T, CH, NW, ALIGNED>(b, jb, row32, sm, int(threadIdx.x));
}

// Rows of two widths in ONE grid of 4-wave workgroups (a T5 block: the 2048-wide rows of q/k/v/o/wi and the
// 5120-wide rows of wo): a workgroup takes four narrow rows, one per wave (no workgroup barriers), or one wide
// row with all four waves.  The wide jobs come first in the table, so their longer workgroups start first and
// the narrow ones fill the tail -- instead of a second, poorly filled launch (2048 workgroups, 1.6 rounds).
template <typename T, int CHW>
__global__ __launch_bounds__(256, 5) void select_rows_mixed_kernel(const SelBatch b) {
    __shared__ union MixedSmem {
        RowSmem<1> narrow[4];
        RowSmem<4> wide;
    } sm;
    uint32_t unit;
    const SelJob &jb = b.job[find_job(b, blockIdx.x, unit)];
    const int tid = threadIdx.x;
    if (jb.in_f > 2048u) {
        select_row_body<T, CHW, 4, true>(b, jb, unit, sm.wide, tid);
    } else {
        const uint32_t row = unit * 4u + uint32_t(tid >> 6);
        if (row < jb.out_f) select_row_body<T, 4, 1, true>(b, jb, row, sm.narrow[tid >> 6], tid & 63);
    }
}

// ------------------------------------------------------------------------------------------
// SEL_MATRIX: thr = sort(score.flatten())[k]; prune score < thr  (wanda_pruner.py:682-683)
//
//   sample   one workgroup per linear: 2048 random samples -> coarse LDS histogram -> a bracket around
//            rank k with a 6-sigma margin (~15 % of the elements); also clears the job's control words
//            and histogram (no memset launches).  One CU issues every sample load, so the sample size
//            is what this kernel costs.
//   count    streams W once: count(key < lo) and the NaN count in registers, keys inside the bracket bump
//            one of 2048 LDS counters ((key-lo) >> shift).  The LAST workgroup to finish (device-scope
//            counter) scans the merged histogram and publishes the bin [lob, lob + 2^shift) that holds
//            rank k and the rank left inside it.
//   apply    streams W again: key < lob -> pruned, key >= lob + 2^shift -> kept, and the few hundred
//            elements inside the bin are kept PROVISIONALLY and appended to a candidate list (index, key).
//            Writes the whole mask, the zeroed weights and the score partial sums: 5 B / weight.
//   resolve  one workgroup per linear: exact rank select among the candidates (LDS radix), then clears
//            the mask byte and the weight of the candidates below the threshold.  The same kernel is the
//            fallback when the bracket missed rank k or the bin overflowed the list (heavy ties, e.g.
//            re-pruning weights that are already half zero): it then selects by streaming the matrix itself.
// Every decision rests on exact counts; sampling only affects speed.  No device-wide fences: everything
// exchanged between workgroups inside a launch goes through device-scope atomics.
// Workspace per job (u32 words): hist[2048] | ctrl[32] | candidates[2 x 8192]  (fused form: the candidate area holds
// the workgroups' key slots [8192] and the histograms of the two refinement levels [2 x 2048]).
// ------------------------------------------------------------------------------------------
constexpr int kMatBins = 2048;
constexpr int kMatSample = 2048;   // one CU issues every sample load: the sample size is what its kernel costs
constexpr int kCandCap = 8192;
constexpr int kCtrl = kMatBins;
constexpr int kCand = kMatBins + 32;
constexpr int kWsWords = kCand + 2 * kCandCap;
enum { C_LO = 0, C_SHIFT, C_BELOW, C_NANC, C_LOB, C_RANKB, C_NONE, C_FAIL, C_DONE, C_NCAND, C_BAR_A, C_BAR_B, C_BAR_A1, C_BAR_A2 };

__device__ __forceinline__ uint32_t ld_dev(const uint32_t *p) {       // device-scope load (bypasses the CU's L1)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Rank search over a histogram held 4 bins per thread by a 1024-thread workgroup (bins past the histogram's
// end are passed as 0): finds the bin with cum <= need < cum + h.  Returns false if need >= total.
__device__ bool block_find_rank(const uint32_t (&h)[4], uint32_t need, uint32_t *red /*>= 20 u32 of LDS*/, uint32_t &bin,
                                uint32_t &before) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t s = h[0] + h[1] + h[2] + h[3];
    const uint32_t incl = wave_incl_scan_u32_dpp(s);
    if (lane == 63) red[wave] = incl;
    if (tid == 0) red[16] = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += red[w];
    const uint32_t hi = off + incl, lo = hi - s;
    if (lo <= need && need < hi) {
        uint32_t cum = lo;
        int i = 0;
        for (; i < 3; ++i) {
            if (cum + h[i] > need) break;
            cum += h[i];
        }
        red[16] = uint32_t(tid * 4 + i);
        red[17] = cum;
        red[18] = h[i];                                          // population of that bin (read by callers that need it)
    }
    __syncthreads();
    bin = red[16];
    before = red[17];
    __syncthreads();
    return bin != 0xFFFFFFFFu;
}

// Two searches over the same histogram in one prefix scan (the sample bracket's two ends): bins of ranks need_a <= need_b.
// A rank >= total gives bin 0xFFFFFFFF.  red: >= 24 words.
__device__ void block_find_rank2(const uint32_t (&h)[4], uint32_t need_a, uint32_t need_b, uint32_t *red, uint32_t &bin_a,
                                 uint32_t &bin_b) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t s = h[0] + h[1] + h[2] + h[3];
    const uint32_t incl = wave_incl_scan_u32_dpp(s);
    if (lane == 63) red[wave] = incl;
    if (tid == 0) { red[16] = 0xFFFFFFFFu; red[20] = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += red[w];
    const uint32_t hi = off + incl, lo = hi - s;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const uint32_t need = which ? need_b : need_a;
        if (lo <= need && need < hi) {
            uint32_t cum = lo;
            int i = 0;
            for (; i < 3; ++i) {
                if (cum + h[i] > need) break;
                cum += h[i];
            }
            red[which ? 20 : 16] = uint32_t(tid * 4 + i);
        }
    }
    __syncthreads();
    bin_a = red[16];
    bin_b = red[20];
    __syncthreads();
}

// Key used inside the streaming passes: score >= +0 or NaN, so clearing the sign bit orders every finite score
// and +inf like score_key() and leaves NaNs above +inf (0x7F800001..0x7FFFFFFF) -- one instruction.  A bin that
// reaches above +inf is handed to the fallback, which uses score_key()'s single NaN key.
__device__ __forceinline__ uint32_t stream_key(float sc) { return __float_as_uint(sc) & 0x7FFFFFFFu; }

template <typename T>
__device__ __forceinline__ uint32_t element_key(const SelJob &jb, uint32_t e) {
    const uint32_t row = e / jb.in_f, col = e - row * jb.in_f;
    const typename T::raw w = static_cast<const typename T::raw *>(jb.W)[int64_t(row) * jb.ldw + col];
    return score_key(ieee_mul(fabsf(to_f32<T>(w)), jb.sq[col]));
}


// The sample of a job: kMatSample elements drawn uniformly with replacement, (row, col) = two multiplicative hashes
// of the sample index scaled by mul-high (no integer division).  `issue` starts the loads, `finish` turns them into
// the bracket [lo, lo + 2048 * 2^shift) around flat rank k (6-sigma margin, resolved to whole coarse bins).
// Every workgroup that runs this on the same W gets the same bracket.
constexpr int kSamplesPerThread = kMatSample / 1024;
template <typename T> struct SampleRegs {
    uint32_t col[kSamplesPerThread];
    typename T::raw wv[kSamplesPerThread];
};
template <typename T>
__device__ __forceinline__ void sample_issue(const SelJob &jb, int tid, SampleRegs<T> &r) {
    const uint32_t numel = jb.out_f * jb.in_f;
    const uint32_t S = numel < uint32_t(kMatSample) ? numel : uint32_t(kMatSample);
    const typename T::raw *W = static_cast<const typename T::raw *>(jb.W);
#pragma unroll
    for (int j = 0; j < kSamplesPerThread; ++j) {
        const uint32_t i = uint32_t(tid) + 1024u * j;
        uint32_t h = (i + 1u) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint32_t row = __umulhi(h, jb.out_f);
        h *= 3266489917u; h ^= h >> 16;
        r.col[j] = __umulhi(h, jb.in_f);
        const bool on = i < S;
        r.col[j] = on ? r.col[j] : 0u;
        r.wv[j] = W[on ? int64_t(row) * jb.ldw + r.col[j] : 0];
    }
}
// hist: kMatBins LDS counters, zeroed and synchronised by the caller; sq: sqrt(scaler_row) (global or LDS)
//
// The bracket [lo, hi] around rank k: the sample's keys at ranks rs -+ 6 sigma.  They are located in a histogram of kMatBins
// EQUAL bins over the sample's own key range [min, max] (about one sample per bin).  (Round 1 binned by sign / exponent / 3
// mantissa bits -- 2^20 keys, an eighth of an octave, per bin -- and snapped the bracket to those edges: it came out two to
// three times as wide as the +-6 sigma it stands for, the bin of rank k then held more keys than the candidate slots take,
// and three of the four linears of a ViT-g block paid for a refinement level: another pass, flush and barrier, ~12 us.)
template <typename T>
__device__ __forceinline__ void sample_finish(const SelJob &jb, int tid, const SampleRegs<T> &r, const float *sq, uint32_t *hist,
                                              uint32_t *red, uint32_t &lo, uint32_t &shift) {
    const uint32_t numel = jb.out_f * jb.in_f;
    const uint32_t S = numel < uint32_t(kMatSample) ? numel : uint32_t(kMatSample);
    uint32_t sk[kSamplesPerThread];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < kSamplesPerThread; ++j) {
        sk[j] = score_key(ieee_mul(fabsf(to_f32<T>(r.wv[j])), sq[r.col[j]]));
        if (uint32_t(tid) + 1024u * j < S) {
            kmin = sk[j] < kmin ? sk[j] : kmin;
            kmax = sk[j] > kmax ? sk[j] : kmax;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t a = uint32_t(__shfl_xor(int(kmin), off, 64)), c = uint32_t(__shfl_xor(int(kmax), off, 64));
        kmin = a < kmin ? a : kmin;
        kmax = c > kmax ? c : kmax;
    }
    __shared__ uint32_t wmin[16], wmax[16];
    if ((tid & 63) == 0) { wmin[tid >> 6] = kmin; wmax[tid >> 6] = kmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        kmin = wmin[w] < kmin ? wmin[w] : kmin;
        kmax = wmax[w] > kmax ? wmax[w] : kmax;
    }
    uint32_t sshift = 0;                                          // sample bins of 2^sshift keys: (kmax - kmin) >> sshift < kMatBins
    while (((kmax - kmin) >> sshift) >= uint32_t(kMatBins)) ++sshift;
#pragma unroll
    for (int j = 0; j < kSamplesPerThread; ++j)
        if (uint32_t(tid) + 1024u * j < S) atomicAdd(&hist[(sk[j] - kmin) >> sshift], 1u);
    __syncthreads();
    const uint32_t rs = uint32_t((uint64_t(jb.k) * S) / numel);
    const float pr = float(jb.k) / float(numel);
    const uint32_t margin = uint32_t(6.f * sqrtf(float(S) * pr * (1.f - pr))) + 8u;
    uint32_t h[4] = {0, 0, 0, 0};
    if (tid < kMatBins / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = hist[tid * 4 + i];
    }
    uint32_t hi = 0xFFFFFFFFu, bin_lo, bin_hi;
    lo = 0;
    block_find_rank2(h, rs > margin ? rs - margin : 0u, rs + margin, red, bin_lo, bin_hi);   // (rank >= S: no bin, open end)
    if (rs > margin && bin_lo != 0xFFFFFFFFu) lo = kmin + (bin_lo << sshift);
    if (bin_hi != 0xFFFFFFFFu) {
        const uint64_t end = uint64_t(kmin) + (uint64_t(bin_hi + 1u) << sshift) - 1u;
        if (end < 0x7F800000ull) hi = uint32_t(end);             // (a bracket reaching Inf / NaN keys stays open above)
    }
    shift = 0;
    while (((hi - lo) >> shift) >= uint32_t(kMatBins)) ++shift;
}

// Four-launch form only: one workgroup per job clears the job's histogram and control words and publishes the bracket.
template <typename T>
__global__ __launch_bounds__(1024) void matrix_sample_kernel(const SelBatch b) {
    __shared__ uint32_t hist[kMatBins];
    __shared__ uint32_t red[24];
    const SelJob &jb = b.job[blockIdx.x];
    const int tid = threadIdx.x;
    uint32_t *ws = jb.ws;
    for (int i = tid; i < kCand; i += 1024) ws[i] = 0;           // histogram + control words
    for (int i = tid; i < kMatBins; i += 1024) hist[i] = 0;
    SampleRegs<T> sr;
    sample_issue<T>(jb, tid, sr);
    __syncthreads();                                             // ctrl words are rewritten below
    uint32_t lo, shift;
    sample_finish<T>(jb, tid, sr, jb.sq, hist, red, lo, shift);
    if (tid == 0) {
        ws[kCtrl + C_LO] = lo;
        ws[kCtrl + C_SHIFT] = shift;
        if (b.p0) ws[kCtrl + C_FAIL] = 1;            // test hook: force the fallback
    }
}

// (row, chunk-in-row) walker over a job's chunks: workgroup `wg` of `nwg`, NCH chunks in flight per lane.
struct ChunkWalk {
    uint32_t cpr, total, step, step_rows, step_cir;
    __device__ ChunkWalk(const SelJob &jb) {
        cpr = (jb.in_f + 7) / 8;
        total = jb.out_f * cpr;
        step = jb.nwg * 1024u;
        step_rows = step / cpr;
        step_cir = step - step_rows * cpr;
    }
};

template <typename T, bool ALIGNED>
__global__ __launch_bounds__(1024) void matrix_count_kernel(const SelBatch b) {
    __shared__ uint32_t lh[kMatBins];
    __shared__ uint32_t red[36];
    uint32_t wg;
    const SelJob &jb = b.job[find_job(b, blockIdx.x, wg)];
    const int tid = threadIdx.x;
    uint32_t *ws = jb.ws;
    // the control block, written by the sample launch: two 16-byte loads issued together
    const uint4 c0 = reinterpret_cast<const uint4 *>(ws + kCtrl)[0], c1 = reinterpret_cast<const uint4 *>(ws + kCtrl)[1];
    if (c1.w /* C_FAIL */) return;
    for (int i = tid; i < kMatBins; i += 1024) lh[i] = 0;
    const uint32_t lo = c0.x /* C_LO */, shift = c0.y /* C_SHIFT */;
    __syncthreads();
    const uint32_t in_f = jb.in_f;
    const typename T::raw *W = static_cast<const typename T::raw *>(jb.W);
    const ChunkWalk cw(jb);
    uint32_t below = 0;
    constexpr int NCH = 4;                                       // chunks in flight per lane
    for (uint32_t cb = wg * 1024u + uint32_t(tid); cb < cw.total; cb += NCH * cw.step) {
        Chunk8<T> raw[NCH];
        float sqv[NCH][8];
        uint32_t col0[NCH];
        bool has[NCH];
        uint32_t row = cb / cw.cpr, cir = cb - row * cw.cpr;     // one division, then carry arithmetic
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            has[u] = cb + uint32_t(u) * cw.step < cw.total;
            col0[u] = cir * 8;
            raw[u] = load_row_chunk<T, ALIGNED>(W + int64_t(has[u] ? row : 0u) * jb.ldw, has[u] ? col0[u] : 0u, in_f);
            row += cw.step_rows;
            cir += cw.step_cir;
            if (cir >= cw.cpr) { cir -= cw.cpr; ++row; }
        }
#pragma unroll
        for (int u = 0; u < NCH; ++u) load_sq_chunk<ALIGNED>(jb.sq, col0[u], in_f, sqv[u]);
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            if (!has[u]) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ALIGNED || col0[u] + j < in_f) {
                    const uint32_t key = stream_key(ieee_mul(fabsf(to_f32<T>(raw[u].v[j])), sqv[u][j]));
                    below += key < lo ? 1u : 0u;
                    const uint32_t d = (key - lo) >> shift;
                    if (key >= lo && d < uint32_t(kMatBins)) atomicAdd(&lh[d], 1u);
                }
            }
        }
    }
    {
        const uint32_t wsum = wave_sum_u32_dpp(below);
        if ((tid & 63) == 0) red[tid >> 6] = wsum;
    }
    __syncthreads();
    for (int i = tid; i < kMatBins; i += 1024) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&ws[i], v);
    }
    if (tid == 0) {
        uint32_t bsum = 0;
        for (int w = 0; w < 16; ++w) bsum += red[w];
        if (bsum) atomicAdd(&ws[kCtrl + C_BELOW], bsum);
    }
    // ---- last workgroup of this job resolves the merged histogram ---------------------------------
    // Everything exchanged between workgroups here goes through device-scope atomics (performed at the
    // memory side, not in the per-XCD L2), so ordering only needs "my atomics have been performed" before
    // the done-counter increment: wait for their acknowledgements.  A __threadfence() would write back and
    // invalidate the XCD's whole L2 in every workgroup (measured: 0.5 ms per launch).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) red[34] = atomicAdd(&ws[kCtrl + C_DONE], 1u);
    __syncthreads();
    if (red[34] != jb.nwg - 1) return;
    uint32_t h[4] = {0, 0, 0, 0};
    if (tid < kMatBins / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = ld_dev(&ws[tid * 4 + i]);
    }
    const uint32_t bel = ld_dev(&ws[kCtrl + C_BELOW]);
    uint32_t bin, before;
    const bool found = block_find_rank(h, jb.k - bel, red, bin, before);   // (wraps harmlessly when bel > k: checked below)
    if (tid == 0) {
        const uint64_t bin_end = uint64_t(lo) + (uint64_t(bin + 1u) << shift);      // one past the bin's last key
        if (bel > jb.k || !found || bin_end > 0x7F800001ull) {
            // the sampled bracket missed rank k, or the bin reaches the NaN keys (a NaN threshold prunes
            // nothing, wanda_pruner.py:683): exact fallback in the resolve kernel
            ws[kCtrl + C_FAIL] = 1;
        } else {
            ws[kCtrl + C_LOB] = lo + (bin << shift);
            ws[kCtrl + C_RANKB] = jb.k - bel - before;
        }
    }
}

template <typename T, bool ALIGNED>
__global__ __launch_bounds__(1024) void matrix_apply_kernel(const SelBatch b) {
    __shared__ double dsm[16];
    uint32_t wg;
    const SelJob &jb = b.job[find_job(b, blockIdx.x, wg)];
    const int tid = threadIdx.x;
    uint32_t *ws = jb.ws;
    const uint4 c0 = reinterpret_cast<const uint4 *>(ws + kCtrl)[0], c1 = reinterpret_cast<const uint4 *>(ws + kCtrl)[1];
    // undecided = nothing is pruned here (fallback: the resolve kernel prunes; NaN threshold: nobody does)
    const bool undecided = c1.w /* C_FAIL */ || c1.z /* C_NONE */;
    const uint32_t shift = c0.y /* C_SHIFT */, lob = c1.x /* C_LOB */;
    // bin = [lob, lob + 2^shift): a one-key bin (shift 0) needs no candidates -- ties with the threshold are kept
    const uint32_t width = shift ? (1u << shift) : 0u;
    const uint32_t in_f = jb.in_f;
    typename T::raw *W = static_cast<typename T::raw *>(jb.W);
    const ChunkWalk cw(jb);
    double dsum = 0.0;
    constexpr int NCH = 2;
    for (uint32_t cb = wg * 1024u + uint32_t(tid); cb < cw.total; cb += NCH * cw.step) {
        Chunk8<T> raw[NCH];
        float sqv[NCH][8];
        uint32_t col0[NCH], rowu[NCH];
        bool has[NCH];
        uint32_t row = cb / cw.cpr, cir = cb - row * cw.cpr;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            has[u] = cb + uint32_t(u) * cw.step < cw.total;
            col0[u] = has[u] ? cir * 8 : 0u;
            rowu[u] = has[u] ? row : 0u;
            raw[u] = load_row_chunk<T, ALIGNED, true>(W + int64_t(rowu[u]) * jb.ldw, col0[u], in_f);
            row += cw.step_rows;
            cir += cw.step_cir;
            if (cir >= cw.cpr) { cir -= cw.cpr; ++row; }
        }
#pragma unroll
        for (int u = 0; u < NCH; ++u) load_sq_chunk<ALIGNED>(jb.sq, col0[u], in_f, sqv[u]);
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            if (!has[u]) continue;
            uint32_t keepbits = 0;
            float fs = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bool pruned = false;
                if (ALIGNED || col0[u] + j < in_f) {
                    const float sc = ieee_mul(fabsf(to_f32<T>(raw[u].v[j])), sqv[u][j]);
                    fs += sc;
                    const uint32_t key = stream_key(sc);
                    pruned = !undecided && key < lob;
                    if (!undecided && key - lob < width && key >= lob) {          // inside the bin: decide later
                        const uint32_t pos = atomicAdd(&ws[kCtrl + C_NCAND], 1u);
                        if (pos < uint32_t(kCandCap)) {
                            ws[kCand + 2 * pos] = rowu[u] * in_f + col0[u] + uint32_t(j);
                            ws[kCand + 2 * pos + 1] = key;
                        }
                    }
                }
                keepbits |= (pruned ? 0u : 1u) << j;
                if (pruned) raw[u].v[j] = typename T::raw(0);
            }
            dsum += double(fs);
            store_mask_chunk<ALIGNED, true>(jb.mask + int64_t(rowu[u]) * in_f, col0[u], in_f, keepbits);
            if (b.apply_zero && keepbits != 0xFFu)
                store_row_chunk<T, ALIGNED, true>(W + int64_t(rowu[u]) * jb.ldw, col0[u], in_f, raw[u]);
        }
    }
    if (jb.parts) {
        dsum = wave_sum_f64(dsum);
        if ((tid & 63) == 0) dsm[tid >> 6] = dsum;
        __syncthreads();
        if (tid == 0) {
            double a = 0.0;
            for (int w = 0; w < 16; ++w) a += dsm[w];
            jb.parts[wg] = a;
        }
        if (wg == 0) {                                           // slots of workgroups this job does not have
            for (uint32_t i = jb.nwg + uint32_t(tid); i < uint32_t(kMatrixParts); i += 1024u) jb.parts[i] = 0.0;
        }
    }
}

// One workgroup per job, after the apply pass: decide the elements the apply pass left undecided.
//   normal    the <= 8192 candidates of the final bin: exact rank select in LDS, prune those below the threshold
//   fallback  the bracket missed rank k (everything is undecided) or the bin overflowed the list: the same
//             select, but streaming the matrix (4 x 8-bit radix over the keys inside [lob, lob + width))
// WRITE_ALL (the fused kernel's own fallback): nothing has been written yet -- select over the whole matrix and
// write EVERY mask byte, without looking at the control words.
// `floor` (WRITE_ALL only): keys below it count as key 0.  The fused kernel hands over a job whose DECIDED chunks are already
// being written -- weights with a key below the undecided bin are turning into zeros while this select makes its passes over
// the matrix, and an element must not be one key in one pass and another in the next.  With every key below the bin's lower
// edge read as 0, the original weight and its zeroed self are the same key in every pass; the threshold (>= the edge) and
// what lies below it are unchanged.
template <typename T, bool WRITE_ALL>
__device__ void matrix_resolve_job(const SelBatch &b, const SelJob &jb, uint32_t *hist, uint32_t *red, uint32_t floor = 0) {
    uint32_t *ws = jb.ws;
    const int tid = threadIdx.x;
    if (!WRITE_ALL && ws[kCtrl + C_NONE]) return;
    const uint32_t fail = WRITE_ALL ? 1u : ws[kCtrl + C_FAIL];
    const uint32_t ncand = WRITE_ALL ? 0u : ws[kCtrl + C_NCAND], shift = WRITE_ALL ? 0u : ws[kCtrl + C_SHIFT];
    const bool from_list = !fail && ncand <= uint32_t(kCandCap);
    if (!fail && (shift == 0 || ncand == 0)) return;            // nothing was left undecided
    const uint32_t lob = fail ? 0u : ws[kCtrl + C_LOB];
    const uint32_t width = fail ? 0xFFFFFFFFu : (1u << shift);  // undecided keys: lob <= key, key - lob < width (or all)
    uint32_t r = fail ? jb.k : ws[kCtrl + C_RANKB];
    const uint32_t numel = jb.out_f * jb.in_f;
    const uint32_t n_items = from_list ? ncand : numel;
    typename T::raw *W = static_cast<typename T::raw *>(jb.W);
    auto key_of = [&](uint32_t i) -> uint32_t {
        if (from_list) return ws[kCand + 2 * i + 1];
        const uint32_t key = element_key<T>(jb, i);
        return key < floor ? 0u : key;
    };
    auto undecided = [&](uint32_t key) -> bool { return fail || (key >= lob && key - lob < width); };
    // rank-r key among the undecided keys: MSD radix, 8 bits per pass
    uint32_t prefix = 0, pmask = 0;
    for (int sh = 24; sh >= 0; sh -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n_items; i += 1024u) {
            const uint32_t key = key_of(i);
            if (undecided(key) && (key & pmask) == prefix) atomicAdd(&hist[(key >> sh) & 255u], 1u);
        }
        __syncthreads();
        uint32_t h[4] = {0, 0, 0, 0};
        if (tid < 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = hist[tid * 4 + i];
        }
        uint32_t bin, before;
        block_find_rank(h, r, red, bin, before);                 // r < number of matching keys by construction
        prefix |= bin << sh;
        pmask |= 0xFFu << sh;
        r -= before;
    }
    const uint32_t thr = prefix;
    const bool none = thr == 0xFFFFFFFFu;                        // NaN threshold (fallback path): prunes nothing
    if (none && !WRITE_ALL) return;
    for (uint32_t i = tid; i < n_items; i += 1024u) {
        const uint32_t key = key_of(i);
        const bool pruned = !none && undecided(key) && key < thr;
        if (pruned || WRITE_ALL) {
            const uint32_t e = from_list ? ws[kCand + 2 * i] : i;
            jb.mask[e] = pruned ? 0 : 1;
            if (pruned && b.apply_zero) {
                const uint32_t row = e / jb.in_f, col = e - row * jb.in_f;
                W[int64_t(row) * jb.ldw + col] = typename T::raw(0);
            }
        }
    }
}

// One workgroup per job, last launch of the four-launch form: decides what the earlier launches left undecided, then
// leaves the job's histogram and control words ZERO -- the state the fused kernel expects to find (include/vlmc.h:
// a SEL_MATRIX workspace is zero-filled when first handed over and is returned zero-filled by every call).
template <typename T>
__global__ __launch_bounds__(1024) void matrix_resolve_kernel(const SelBatch b) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t red[24];
    const SelJob &jb = b.job[blockIdx.x];
    matrix_resolve_job<T, false>(b, jb, hist, red);
    __syncthreads();
#ifdef VLMC_FUSED_STAMPS
    for (int i = threadIdx.x; i < kCtrl + 16; i += 1024) jb.ws[i] = 0;       // (keep the phase clocks)
#else
    for (int i = threadIdx.x; i < kCand; i += 1024) jb.ws[i] = 0;
#endif
    // (the candidate pairs of this form overlap the fused kernel's refinement histograms, which it expects zeroed)
    for (int i = threadIdx.x; i < 2 * kMatBins; i += 1024) jb.ws[kCand + 12288 + i] = 0;      // (= kHist1 of the fused kernel)
}

// ------------------------------------------------------------------------------------------
// SEL_MATRIX, fused form (default): count + candidates + apply in ONE launch of co-resident workgroups.
//
// The weights of a whole ViT-g block (25.2 M x 2 B = 50.5 MB) fit in the register files of the chip
// (256 CUs x 512 KB): every lane loads its R 16-byte chunks ONCE, keeps them in VGPRs, and the threshold
// is agreed on through two grid barriers per linear, so W is read once and mask / zeroed W written once
// -- exactly the 5 B / weight of SURVEY 8(d) instead of 7 B, and one launch instead of three.
//   P0  every workgroup draws the job's 2048-element sample itself (same hashes => same bracket [lo, lo + 2048 * 2^s)
//       everywhere, nothing to exchange); the sample's loads are issued before the row chunks, its histogram and
//       rank search run while the chunks stream in
//   P1  load R chunks per lane (chunks past R x grid are streamed and re-read in the later phases);
//       count(key < lo), 2048-bin LDS histogram of the sampled bracket, score partial sums; flush the
//       histogram with device-scope atomics                                          -- barrier A
//   P2  every workgroup scans the merged histogram itself (same data => same decision everywhere): the bin
//       [lob, lob + 2^shift) that holds rank k; lanes append the KEYS of their elements inside it (a few
//       hundred per linear) to the candidate list                                    -- barrier B
//   P3  every workgroup radix-selects the exact threshold among the candidates in LDS
//   P4  apply from the registers: one compare per key -> mask bytes + zeroed weights.
// Barriers: one arrival counter per barrier in the control block, device-scope atomics only (no L2
// write-back fences, see matrix_count_kernel).  The grid never exceeds one workgroup per CU, so all
// workgroups are resident on an otherwise idle GPU; if they are not (CUs held by another stream), the
// bounded spin runs out and a waiter raises the barrier's fail bit: then NO workgroup of the job passes that
// barrier, nobody touches W or the mask, and the job's last workgroup to finish (done counter) does the exact
// streaming select over the whole matrix (matrix_resolve_job<WRITE_ALL>) -- as it does when the sampled bracket
// missed rank k, the bin reaches the NaN keys or a skewed bin overflows one workgroup's slot.  Every wave therefore
// reaches the end of the kernel whatever the residency.
// The job's global histogram and control words must be ZERO on entry; the last workgroup zeroes them again.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kBarFail = 0x80000000u;
constexpr int kFusedCand = 4096;                 // candidate keys a workgroup can hold (more => exact fallback)
constexpr int kSlotArea = 12288;                 // words of the workspace's candidate area used for the slots
constexpr int kSlotMax = 128;                    // candidate keys one workgroup may publish: min(128, kSlotArea / workgroups)
constexpr int kMaxFusedWgs = 256;                // => at least 48 keys per slot
constexpr int kHist1 = kCand + kSlotArea;        // global histograms of the refinement levels 1 and 2 (2 x 2048 words)
constexpr uint32_t kSlotEmpty = 0xFFFFFFFFu, kSlotOverflow = 0xFFFFFFFEu;   // (keys are relative to the bin: < 2^22)
constexpr int kFusedMaxIn = 8192;                // in_features the LDS copy of sqrt(scaler_row) can hold
constexpr uint32_t kSpinMax = 1u << 12;          // x (device-scope load + s_sleep, ~2.2 us) ~ 9 ms; a barrier normally takes < 20 us

// tid 0 of a workgroup: arrive at the barrier word and wait for `n` arrivals.  false = the barrier failed.
// The fail bit can only be set (compare-and-swap) while the count is still short, and whoever arrives or polls
// afterwards sees it before the count: either every workgroup of the job passes the barrier or none does.
__device__ __forceinline__ bool grid_arrive_wait(uint32_t *ws, int which, uint32_t n) {
    uint32_t *ctr = ws + kCtrl + which;
    uint32_t v = atomicAdd(ctr, 1u) + 1u;
    for (uint32_t it = 0;; ++it) {
        if (v & kBarFail) return false;
        if (v >= n) return true;
        if (it >= kSpinMax) {                                    // give up: declare the failure unless it completed meanwhile
            const uint32_t seen = atomicCAS(ctr, v, v | kBarFail);
            if (seen == v) return false;
            v = seen;
            continue;
        }
        __builtin_amdgcn_s_sleep(8);
        v = ld_dev(ctr);
    }
}

#ifdef VLMC_FUSED_STAMPS
// diagnostic build only (never shipped): 100 MHz wall-clock stamps of workgroup 0 / the last workgroup of each job
#define VLMC_FSTAMP(i)                                                                               \
    do {                                                                                             \
        if (tid == 0 && (wg == 0 || wg == jb.nwg - 1))                                               \
            ws[kCtrl + 16 + (wg == 0 ? 0 : 8) + 2 * 0 + (i)] = uint32_t(wall_clock64());             \
    } while (0)
#else
#define VLMC_FSTAMP(i) do {} while (0)
#endif

template <typename T, bool ALIGNED, int R>
__global__ __launch_bounds__(1024, 1) void matrix_fused_kernel(const SelBatch b) {
    __shared__ uint32_t lh[kMatBins];
    __shared__ uint32_t cand[kFusedCand];
    __shared__ float sqs[kFusedMaxIn];                           // sqrt(scaler_row): read 3 x per element
    __shared__ uint32_t red[40];
    __shared__ double dsm[16];
    uint32_t wg;
    const SelJob &jb = b.job[find_job(b, blockIdx.x, wg)];
    const int tid = threadIdx.x;
    uint32_t *ws = jb.ws;
    bool fail = b.p0 == 1;                                       // test hook: force the fallback
    const uint32_t in_f = jb.in_f;
    typename T::raw *W = static_cast<typename T::raw *>(jb.W);
    // ---- P0: the sample's loads go out first, its bracket is computed while the row chunks stream in ----------
    SampleRegs<T> sr;
    sample_issue<T>(jb, tid, sr);
    for (uint32_t i = tid; i < in_f; i += 1024u) sqs[i] = jb.sq[i];
    const ChunkWalk cw(jb);
    const uint32_t cb0 = wg * 1024u + uint32_t(tid);
    const uint32_t row0 = cb0 / cw.cpr, cir0 = cb0 - row0 * cw.cpr;
    VLMC_FSTAMP(0);
#define VLMC_WALK_NEXT(row, cir) \
    do { row += cw.step_rows; cir += cw.step_cir; if (cir >= cw.cpr) { cir -= cw.cpr; ++row; } } while (0)

    // ---- P1: load, count ------------------------------------------------------------------------------
    Chunk8<T> raw[R];
    {
        uint32_t row = row0, cir = cir0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (cb0 + uint32_t(u) * cw.step < cw.total)
                raw[u] = load_row_chunk<T, ALIGNED, true>(W + int64_t(row) * jb.ldw, cir * 8, in_f);
            VLMC_WALK_NEXT(row, cir);
        }
    }
    for (int i = tid; i < kMatBins; i += 1024) lh[i] = 0;
    __syncthreads();
    uint32_t lo, bshift;
    sample_finish<T>(jb, tid, sr, sqs, lh, red, lo, bshift);     // (ends with a workgroup barrier)
    for (int i = tid; i < kMatBins; i += 1024) lh[i] = 0;
    __syncthreads();
    auto load_sq = [&](uint32_t col0, float *o) {
        if constexpr (ALIGNED) {
            const float4 a = reinterpret_cast<const float4 *>(sqs + col0)[0], c = reinterpret_cast<const float4 *>(sqs + col0)[1];
            o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (col0 + j < in_f) ? sqs[col0 + j] : 0.f;
        }
    };
    uint32_t below = 0;
    double dsum = 0.0;
    auto count_chunk = [&](const Chunk8<T> &c, uint32_t col0) {
        float sq[8];
        load_sq(col0, sq);
        float fs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (ALIGNED || col0 + j < in_f) {
                const float sc = ieee_mul(fabsf(to_f32<T>(c.v[j])), sq[j]);
                fs += sc;
                const uint32_t key = stream_key(sc);
                below += key < lo ? 1u : 0u;
                const uint32_t d = (key - lo) >> bshift;
                if (key >= lo && d < uint32_t(kMatBins)) atomicAdd(&lh[d], 1u);
            }
        }
        dsum += double(fs);
    };
    {
        uint32_t row = row0, cir = cir0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (cb0 + uint32_t(u) * cw.step < cw.total) count_chunk(raw[u], cir * 8);
            VLMC_WALK_NEXT(row, cir);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (uint32_t cb = cb0 + uint32_t(R) * cw.step; cb < cw.total; cb += cw.step) {       // past the registers
            count_chunk(load_row_chunk<T, ALIGNED>(W + int64_t(row) * jb.ldw, cir * 8, in_f), cir * 8);
            VLMC_WALK_NEXT(row, cir);
        }
    }
    VLMC_FSTAMP(1);
    uint32_t thr = 0, deferred = 0, early_floor = 0;
    bool early = false;                                          // decided register chunks were written before barrier B
    if (!fail) {
        const uint32_t wsum = wave_sum_u32_dpp(below);
        if ((tid & 63) == 0) red[tid >> 6] = wsum;
        __syncthreads();
        for (int i = tid; i < kMatBins / 2; i += 1024) {         // two bins per 64-bit atomic (no carry: counts < 2^32)
            const unsigned long long v = (unsigned long long)(lh[2 * i]) | ((unsigned long long)(lh[2 * i + 1]) << 32);
            if (v) atomicAdd(reinterpret_cast<unsigned long long *>(ws) + i, v);
        }
        if (tid == 0) {
            uint32_t bsum = 0;
            for (int w = 0; w < 16; ++w) bsum += red[w];
            if (bsum) atomicAdd(&ws[kCtrl + C_BELOW], bsum);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my atomics have been performed
        __syncthreads();
#ifndef VLMC_FUSED_STAMP_SCAN
        VLMC_FSTAMP(2);
#endif
        if (tid == 0) red[34] = grid_arrive_wait(ws, C_BAR_A, jb.nwg) ? 1u : 0u;      // ---- barrier A
        __syncthreads();
        fail = red[34] == 0;
        VLMC_FSTAMP(3);
    }
    // ---- P2: the bin of rank k; candidates ----------------------------------------------------------------
    if (!fail) {
        uint32_t h[4] = {0, 0, 0, 0};
        if (tid < kMatBins / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = ld_dev(&ws[tid * 4 + i]);
        }
        const uint32_t bel = ld_dev(&ws[kCtrl + C_BELOW]);
        uint32_t bin, before;
        const bool found = block_find_rank(h, jb.k - bel, red, bin, before);
        const uint64_t bin_end = uint64_t(lo) + (uint64_t(bin + 1u) << bshift);
        // the sampled bracket missed rank k, or the bin reaches the NaN keys: exact fallback (the job's last workgroup)
        fail = bel > jb.k || !found || bin_end > 0x7F800001ull;
        if (!fail) {
            uint32_t lob = lo + (bin << bshift), rankb = jb.k - bel - before, pop = red[18];
#ifdef VLMC_FUSED_STAMP_SCAN                                      // (diagnostic: see the stamp below)
            const uint32_t dbg_pop0 = pop;
            uint32_t dbg_levels = 0;
#endif
            const uint32_t slot_cap = min(uint32_t(kSlotMax), uint32_t(kSlotArea) / jb.nwg);
            // ---- refinement: a bin too crowded for the candidate slots (ties: re-pruning weights that are already half
            // ---- zero, dead input channels) is histogrammed again, 11 more key bits per level, from the registers; after
            // ---- at most two levels a bin is ONE key value and needs no candidates at all (ties with the threshold stay)
            for (int level = 1; level <= 2 && !fail && bshift != 0 && (pop > (slot_cap / 4u) * jb.nwg || pop > 2048u); ++level) {
                const uint32_t nshift = bshift > 11u ? bshift - 11u : 0u, base = lob, width = 1u << bshift;
#ifdef VLMC_FUSED_STAMP_SCAN
                ++dbg_levels;
#endif
                uint32_t *gh = ws + kHist1 + (level - 1) * kMatBins;
                for (int i = tid; i < kMatBins; i += 1024) lh[i] = 0;
                __syncthreads();
                auto bin_chunk = [&](const Chunk8<T> &c, uint32_t col0) {
                    float sq[8];
                    load_sq(col0, sq);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ALIGNED || col0 + j < in_f) {
                            const uint32_t key = stream_key(ieee_mul(fabsf(to_f32<T>(c.v[j])), sq[j]));
                            if (key >= base && key - base < width) atomicAdd(&lh[(key - base) >> nshift], 1u);
                        }
                    }
                };
                uint32_t row = row0, cir = cir0;
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    if (cb0 + uint32_t(u) * cw.step < cw.total) bin_chunk(raw[u], cir * 8);
                    VLMC_WALK_NEXT(row, cir);
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (uint32_t cb = cb0 + uint32_t(R) * cw.step; cb < cw.total; cb += cw.step) {
                    bin_chunk(load_row_chunk<T, ALIGNED>(W + int64_t(row) * jb.ldw, cir * 8, in_f), cir * 8);
                    VLMC_WALK_NEXT(row, cir);
                }
                __syncthreads();
                for (int i = tid; i < kMatBins / 2; i += 1024) {
                    const unsigned long long v = (unsigned long long)(lh[2 * i]) | ((unsigned long long)(lh[2 * i + 1]) << 32);
                    if (v) atomicAdd(reinterpret_cast<unsigned long long *>(gh) + i, v);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) red[34] = grid_arrive_wait(ws, level == 1 ? C_BAR_A1 : C_BAR_A2, jb.nwg) ? 1u : 0u;
                __syncthreads();
                fail = red[34] == 0;
                if (!fail) {
                    uint32_t hh[4] = {0, 0, 0, 0};
                    if (tid < kMatBins / 4) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) hh[i] = ld_dev(&gh[tid * 4 + i]);
                    }
                    uint32_t bn, bf;
                    if (!block_find_rank(hh, rankb, red, bn, bf)) fail = true;       // (cannot happen: rankb < pop)
                    lob = base + (bn << nshift);
                    rankb -= bf;
                    pop = red[18];
                    bshift = nshift;
                    __syncthreads();
                }
            }
            if (fail) bshift = 0;
            // the slot a workgroup publishes its candidates in: sized by the bin's population (the same number in every
            // workgroup), 4 x its share + 16 -- P3 loads every slot of the job whatever they hold
            const uint32_t slot = min(slot_cap, max(32u, 4u * (pop / jb.nwg) + 16u));
            thr = lob;                                           // one-key bin: ties with the threshold are kept
#ifdef VLMC_FUSED_STAMP_SCAN         // diagnostic: the "flush" stamp marks the end of the scan (refinement levels included) and
            VLMC_FSTAMP(2);          // the "P1 count" slot shows keys in the first bin / 100 + 10 000 x refinement levels
            if (tid == 0 && (wg == 0 || wg == jb.nwg - 1))
                ws[kCtrl + 16 + (wg == 0 ? 0 : 8) + 1] = ws[kCtrl + 16 + (wg == 0 ? 0 : 8) + 0] + dbg_pop0 + 1000000u * dbg_levels;
#endif
            if (bshift) {
                const uint32_t width = 1u << bshift;
                if (tid == 0) red[35] = 0;
                __syncthreads();
                // ONE pass over the registers does both jobs.  A chunk with no key inside the undecided bin [lob, lob + width)
                // is decided by the merged histogram alone -- key < lob: pruned, anything else kept, whatever the exact
                // threshold inside the bin turns out to be -- and is written NOW: the 76 MB of stores drain while the
                // candidates are exchanged (barrier B) and ranked (P3).  The few hundred chunks that do hold such a key
                // give their keys to the candidate list and wait for the threshold (`deferred`, one bit per register chunk).
                // (Before: a candidate pass with a branch per element, 19 us, and the stores only after P3.)
                // A job that fails from here on has part of W and of the mask written (and still being written): every such
                // element is on its final side of any threshold inside the bin; the last workgroup's exact select over the
                // whole matrix reads every key below the bin as 0 (matrix_resolve_job's `floor`), so a weight and its zeroed
                // self are one key to it, and it decides -- and writes -- the same.
                early = true;
                early_floor = lob;
                auto early_chunk = [&](Chunk8<T> &c, uint32_t rowu, uint32_t col0) -> bool {
                    float sq[8];
                    load_sq(col0, sq);
                    uint32_t key[8];
                    uint32_t keepbits = 0, inb = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ALIGNED || col0 + j < in_f) {
                            key[j] = stream_key(ieee_mul(fabsf(to_f32<T>(c.v[j])), sq[j]));
                            keepbits |= (key[j] >= lob ? 1u : 0u) << j;
                            inb |= ((key[j] >= lob && key[j] - lob < width) ? 1u : 0u) << j;
                        } else {
                            key[j] = 0;
                            keepbits |= 1u << j;
                        }
                    }
                    if (inb) {                                               // rare: one branch per chunk
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if ((inb >> j) & 1u) {
                                const uint32_t pos = atomicAdd(&red[35], 1u);
                                if (pos < uint32_t(kFusedCand)) cand[pos] = key[j] - lob;
                            }
                        return true;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (!((keepbits >> j) & 1u)) c.v[j] = typename T::raw(0);
#ifdef VLMC_FUSED_EXP                                             // diagnostic builds only (never shipped): which stores cost what
                    if (!(VLMC_FUSED_EXP & 2)) store_mask_chunk<ALIGNED, !(VLMC_FUSED_EXP & 4)>(jb.mask + int64_t(rowu) * in_f, col0, in_f, keepbits);
                    if (!(VLMC_FUSED_EXP & 1) && b.apply_zero && keepbits != 0xFFu)
                        store_row_chunk<T, ALIGNED, !(VLMC_FUSED_EXP & 4)>(W + int64_t(rowu) * jb.ldw, col0, in_f, c);
#else
                    store_mask_chunk<ALIGNED, true>(jb.mask + int64_t(rowu) * in_f, col0, in_f, keepbits);
                    if (b.apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED, true>(W + int64_t(rowu) * jb.ldw, col0, in_f, c);
#endif
                    return false;
                };
                auto cand_chunk = [&](const Chunk8<T> &c, uint32_t col0) {
                    float sq[8];
                    load_sq(col0, sq);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ALIGNED || col0 + j < in_f) {
                            const uint32_t key = stream_key(ieee_mul(fabsf(to_f32<T>(c.v[j])), sq[j]));
                            if (key >= lob && key - lob < width) {               // collected in LDS first
                                const uint32_t pos = atomicAdd(&red[35], 1u);
                                if (pos < uint32_t(kFusedCand)) cand[pos] = key - lob;
                            }
                        }
                    }
                };
                uint32_t row = row0, cir = cir0;
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    if (cb0 + uint32_t(u) * cw.step < cw.total && early_chunk(raw[u], row, cir * 8)) deferred |= 1u << u;
                    VLMC_WALK_NEXT(row, cir);
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (uint32_t cb = cb0 + uint32_t(R) * cw.step; cb < cw.total; cb += cw.step) {   // past the registers: keys only
                    cand_chunk(load_row_chunk<T, ALIGNED>(W + int64_t(row) * jb.ldw, cir * 8, in_f), cir * 8);
                    VLMC_WALK_NEXT(row, cir);
                }
                __syncthreads();
                // every workgroup publishes its (few) candidates in its OWN slot of the list, padded with a sentinel:
                // no reservation round trip, and the readers need no count before they can issue their loads.
                // (A per-element atomic on one shared counter serialises at the memory side: ~30 ns x 600 per linear.)
                if (uint32_t(tid) < slot) {
                    const uint32_t mine = red[35];
                    const uint32_t v = mine > slot ? kSlotOverflow : (uint32_t(tid) < mine ? cand[tid] : kSlotEmpty);
                    __hip_atomic_store(&ws[kCand + wg * slot + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                VLMC_FSTAMP(4);
                if (tid == 0) red[34] = grid_arrive_wait(ws, C_BAR_B, jb.nwg) ? 1u : 0u;     // ---- barrier B
                __syncthreads();
                fail = red[34] == 0 || b.p0 == 2;                // (test hook 2: fail with the decided chunks already written)
                VLMC_FSTAMP(5);
                // ---- P3: exact rank `rankb` among the candidates (keys relative to lob, < 2^bshift) ----------
                uint32_t ncand = 0;
                if (!fail) {
                    if (tid == 0) { red[35] = 0; red[38] = 0; }
                    __syncthreads();
                    for (uint32_t i = tid; i < jb.nwg * slot; i += 1024u) {                // all slots, one round trip
                        const uint32_t v = ld_dev(&ws[kCand + i]);
                        if (v == kSlotOverflow) red[38] = 1;
                        else if (v != kSlotEmpty) {
                            const uint32_t pos = atomicAdd(&red[35], 1u);
                            if (pos < uint32_t(kFusedCand)) cand[pos] = v;
                        }
                    }
                    __syncthreads();
                    ncand = red[35];
                    if (red[38] || ncand > uint32_t(kFusedCand)) fail = true;   // heavy ties: more than a slot / the list holds
                }
                if (!fail) {
                    uint32_t prefix = 0, pmask = 0, r = rankb;
                    for (int sh = int((bshift - 1u) & ~7u); sh >= 0; sh -= 8) {
                        if (tid < 256) lh[tid] = 0;
                        __syncthreads();                          // (also: cand[] complete)
                        for (uint32_t i = tid; i < ncand; i += 1024u) {
                            const uint32_t key = cand[i];
                            if ((key & pmask) == prefix) atomicAdd(&lh[(key >> sh) & 255u], 1u);
                        }
                        __syncthreads();
                        uint32_t hh[4] = {0, 0, 0, 0};
                        if (tid < 64) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) hh[i] = lh[tid * 4 + i];
                        }
                        uint32_t bn, bf;
                        block_find_rank(hh, r, red, bn, bf);      // r < matching keys by construction
                        prefix |= bn << sh;
                        pmask |= 0xFFu << sh;
                        r -= bf;
                    }
                    thr = lob + prefix;
                }
            }
        }
    }
    VLMC_FSTAMP(6);
    // ---- P4: apply --------------------------------------------------------------------------------------
    // my last look at the workspace is behind me: the done count (its round trip hides behind the stores)
    if (tid == 0) red[37] = atomicAdd(&ws[kCtrl + C_DONE], 1u);
    auto apply_chunk = [&](Chunk8<T> &c, uint32_t rowu, uint32_t col0) {
        uint32_t keepbits = 0;
        float sq[8];
        load_sq(col0, sq);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bool pruned = false;
            if (ALIGNED || col0 + j < in_f)
                pruned = stream_key(ieee_mul(fabsf(to_f32<T>(c.v[j])), sq[j])) < thr;
            keepbits |= (pruned ? 0u : 1u) << j;
            if (pruned) c.v[j] = typename T::raw(0);
        }
#ifdef VLMC_FUSED_EXP                                             // diagnostic builds only (never shipped): which stores cost what
        if (!(VLMC_FUSED_EXP & 2)) store_mask_chunk<ALIGNED, !(VLMC_FUSED_EXP & 4)>(jb.mask + int64_t(rowu) * in_f, col0, in_f, keepbits);
        if (!(VLMC_FUSED_EXP & 1) && b.apply_zero && keepbits != 0xFFu)
            store_row_chunk<T, ALIGNED, !(VLMC_FUSED_EXP & 4)>(W + int64_t(rowu) * jb.ldw, col0, in_f, c);
#else
        store_mask_chunk<ALIGNED, true>(jb.mask + int64_t(rowu) * in_f, col0, in_f, keepbits);
        if (b.apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED, true>(W + int64_t(rowu) * jb.ldw, col0, in_f, c);
#endif
    };
    if (!fail) {                                                 // (a failed job is written by its last workgroup, below)
        uint32_t row = row0, cir = cir0;
#pragma unroll
        for (int u = 0; u < R; ++u) {                            // (after an early pass: only the chunks that waited for thr)
            if (cb0 + uint32_t(u) * cw.step < cw.total && (!early || ((deferred >> u) & 1u))) apply_chunk(raw[u], row, cir * 8);
            VLMC_WALK_NEXT(row, cir);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (uint32_t cb = cb0 + uint32_t(R) * cw.step; cb < cw.total; cb += cw.step) {
            Chunk8<T> c = load_row_chunk<T, ALIGNED, true>(W + int64_t(row) * jb.ldw, cir * 8, in_f);
            apply_chunk(c, row, cir * 8);
            VLMC_WALK_NEXT(row, cir);
        }
    }
#undef VLMC_WALK_NEXT
    VLMC_FSTAMP(7);
    if (jb.parts) {
        dsum = wave_sum_f64(dsum);
        if ((tid & 63) == 0) dsm[tid >> 6] = dsum;
        __syncthreads();
        if (tid == 0) {
            double a = 0.0;
            for (int w = 0; w < 16; ++w) a += dsm[w];
            jb.parts[wg] = a;
        }
        if (wg == 0) {
            for (uint32_t i = jb.nwg + uint32_t(tid); i < uint32_t(kMatrixParts); i += 1024u) jb.parts[i] = 0.0;
        }
    }
    // ---- the job's last workgroup: exact fallback of a failed job (every workgroup of the job knows `fail`, and none
    // ---- of them has written W or the mask), then the histogram and control words go back to zero -------------------
    __syncthreads();
    if (red[37] != jb.nwg - 1) return;
    if (fail) matrix_resolve_job<T, true>(b, jb, lh, red, early ? early_floor : 0u);
    __syncthreads();
#ifdef VLMC_FUSED_STAMPS
    for (int i = tid; i < kCtrl + 16; i += 1024)                 // (keep the phase clocks)
#else
    for (int i = tid; i < kCand; i += 1024)
#endif
        __hip_atomic_store(&ws[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < 2 * kMatBins; i += 1024)               // the refinement levels' histograms
        __hip_atomic_store(&ws[kHist1 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// SEL_NM
// ------------------------------------------------------------------------------------------
template <typename T, bool ALIGNED, int M>
__global__ __launch_bounds__(256) void nm_kernel(const SelBatch b) {
    __shared__ double dsm[4];
    uint32_t wg;
    const SelJob &jb = b.job[find_job(b, blockIdx.x, wg)];
    const int n = int(b.p0);
    const uint32_t in_f = jb.in_f, cpr = (in_f + 7) / 8, total = jb.out_f * cpr;
    typename T::raw *W = static_cast<typename T::raw *>(jb.W);
    double dsum = 0.0;
    for (uint32_t c = wg * 256u + threadIdx.x; c < total; c += jb.nwg * 256u) {
        const uint32_t row = c / cpr, col0 = (c - row * cpr) * 8;
        typename T::raw *wrow = W + int64_t(row) * jb.ldw;
        Chunk8<T> raw = load_row_chunk<T, ALIGNED, true>(wrow, col0, in_f);
        float sqv[8];
        load_sq_chunk<ALIGNED>(jb.sq, col0, in_f, sqv);
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
            uint32_t prunedbits = 0, at_n = 0, at_n1 = 0;                    // keys of stable ranks n - 1 and n
#pragma unroll
            for (int i = 0; i < M; ++i) {
                // stable rank of column i inside its group: smaller key first, then lower index
                int rank = 0;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const uint32_t kj = key[g * M + j], ki = key[g * M + i];
                    rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
                }
                prunedbits |= (rank < n ? 1u : 0u) << i;
                at_n = rank == n - 1 ? key[g * M + i] : at_n;
                at_n1 = rank == n ? key[g * M + i] : at_n1;
            }
            // a tie across the selection boundary: the columns `torch.topk` returns on the CPU (topk_order.hpp)
            if (M > 2 && n > 0 && n < M && at_n == at_n1) {
                uint32_t gk[M];
#pragma unroll
                for (int i = 0; i < M; ++i) gk[i] = key[g * M + i];
                prunedbits = torch_cpu_smallest<M>(gk, n);
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const bool pruned = ((prunedbits >> i) & 1u) && (ALIGNED || col0 + g * M + i < in_f);
                keepbits |= (pruned ? 0u : 1u) << (g * M + i);
                if (pruned) raw.v[g * M + i] = typename T::raw(0);
            }
        }
        store_mask_chunk<ALIGNED, true>(jb.mask + int64_t(row) * in_f, col0, in_f, keepbits);
        if (b.apply_zero && keepbits != 0xFFu) store_row_chunk<T, ALIGNED, true>(wrow, col0, in_f, raw);
    }
    if (jb.parts) {
        dsum = wave_sum_f64(dsum);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) dsm[wave] = dsum;
        __syncthreads();
        if (threadIdx.x == 0) jb.parts[wg] = dsm[0] + dsm[1] + dsm[2] + dsm[3];
        if (wg == 0) {
            for (uint32_t i = jb.nwg + threadIdx.x; i < uint32_t(kNmParts); i += 256u) jb.parts[i] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static size_t ws_bytes(int mode) { return mode == VLMC_SEL_MATRIX ? round_up(size_t(kWsWords) * 4, 256) : 0; }
static int64_t n_partials(int mode, int64_t out_f) {
    return mode == VLMC_SEL_ROW ? out_f : (mode == VLMC_SEL_MATRIX ? kMatrixParts : kNmParts);
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

static bool job_aligned(const vlmc_select_job &j) {
    return j.in_features % 8 == 0 && j.ldw % 8 == 0 && aligned16(j.W) && aligned16(j.sqrt_scaler) &&
           (reinterpret_cast<uintptr_t>(j.mask) % 8) == 0;
}

static void fill_job(SelJob &d, const vlmc_select_job &j) {
    d.W = j.W; d.sq = j.sqrt_scaler; d.mask = j.mask; d.parts = j.score_partials;
    d.ws = static_cast<uint32_t *>(j.workspace);
    d.out_f = uint32_t(j.out_features); d.in_f = uint32_t(j.in_features); d.ldw = uint32_t(j.ldw);
    d.k = uint32_t(j.k); d.unit_end = 0; d.nwg = 0;
}

// (NW waves) x (CH chunks of 8 columns per lane) must cover the row: 64*NW*CH >= in/8.
// Few rows -> more waves per row (parallelism); many rows -> one wave per row (no barriers).
template <typename T>
static int launch_rows(const vlmc_select_job *jobs, const int *idx, int n, int apply_zero, hipStream_t st) {
    const vlmc_select_job &j0 = jobs[idx[0]];
    const int64_t in_f = j0.in_features, nchunks = (in_f + 7) / 8;
    if (nchunks > 2048) {
        set_error("vlmc_wanda_select: in_features %lld too large (max 16384)", (long long)in_f);
        return VLMC_EINVAL;
    }
    SelBatch b;
    b.n = n; b.apply_zero = apply_zero;
    b.p0 = uint32_t(env_int("VLMC_SELECT_SAMPLE_MARGIN", 9));          // ~ +-3 sigma of a 32-sample rank
    b.p1 = 0;
    int64_t rows = 0;
    for (int i = 0; i < n; ++i) {
        fill_job(b.job[i], jobs[idx[i]]);
        rows += jobs[idx[i]].out_features;
        b.job[i].unit_end = uint32_t(rows);
        b.job[i].nwg = uint32_t((uint64_t(jobs[idx[i]].k) << 16) / uint64_t(in_f));
    }
    const bool aligned = job_aligned(j0);
#define VLMC_ROWS(CH, NW, AL) \
    VLMC_LAUNCH_TIMED((select_rows_kernel<T, CH, NW, AL>), dim3(unsigned(rows)), dim3(64 * NW), st, b)
    if (!aligned) {
        if (nchunks <= 256) VLMC_ROWS(4, 1, false);
        else VLMC_ROWS(4, 8, false);
        return VLMC_OK;
    }
    int nw = 1;
    while (nw < 8 && nchunks > int64_t(64) * nw * 4) nw *= 2;               // CH <= 4
    const int64_t want_waves = env_int("VLMC_SELECT_WANT_WAVES", 4096);      // ~4 waves per SIMD
    while (nw < 8 && rows * nw < want_waves && nchunks > int64_t(64) * nw) nw *= 2;
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

// All SEL_ROW jobs of a call in one mixed launch: possible when every job is 16-byte aligned, every row fits
// 4 waves x 4 chunks (in <= 8192) and both widths occur (one width alone: launch_rows is the tuned path).
static bool mixed_rows_eligible(const vlmc_select_job *jobs, int n_jobs) {
    if (n_jobs < 2 || n_jobs > kMaxSelJobs || !env_int("VLMC_SELECT_MIXED", 1)) return false;
    bool narrow = false, wide = false;
    for (int i = 0; i < n_jobs; ++i) {
        if (!job_aligned(jobs[i]) || jobs[i].in_features > 8192) return false;
        (jobs[i].in_features > 2048 ? wide : narrow) = true;
    }
    return narrow && wide;
}

template <typename T>
static int launch_rows_mixed(const vlmc_select_job *jobs, int n_jobs, int apply_zero, hipStream_t st) {
    SelBatch b;
    b.n = n_jobs; b.apply_zero = apply_zero;
    b.p0 = uint32_t(env_int("VLMC_SELECT_SAMPLE_MARGIN", 9));
    b.p1 = 0;
    uint32_t units = 0;
    int64_t max_chunks = 0;
    int n = 0;
    const bool wide_first = env_int("VLMC_MIXED_WIDE_FIRST", 1) != 0;   // (0: narrow first -- measured 62.5 vs 58.9 us per T5 block)
    for (int pass = 0; pass < 2; ++pass) {                      // wide jobs first
        for (int i = 0; i < n_jobs; ++i) {
            const vlmc_select_job &j = jobs[i];
            const bool is_wide = j.in_features > 2048;
            if (is_wide != (pass == (wide_first ? 0 : 1))) continue;
            fill_job(b.job[n], j);
            units += uint32_t(is_wide ? j.out_features : (j.out_features + 3) / 4);
            b.job[n].unit_end = units;
            b.job[n].nwg = uint32_t((uint64_t(j.k) << 16) / uint64_t(j.in_features));
            if (is_wide && (j.in_features + 7) / 8 > max_chunks) max_chunks = (j.in_features + 7) / 8;
            ++n;
        }
    }
    const int chw = int((max_chunks + 255) / 256);
    switch (chw) {
        case 2: VLMC_LAUNCH_TIMED((select_rows_mixed_kernel<T, 2>), dim3(units), dim3(256), st, b); break;
        case 3: VLMC_LAUNCH_TIMED((select_rows_mixed_kernel<T, 3>), dim3(units), dim3(256), st, b); break;
        default: VLMC_LAUNCH_TIMED((select_rows_mixed_kernel<T, 4>), dim3(units), dim3(256), st, b); break;
    }
    return VLMC_OK;
}

// elementwise passes: workgroups of `threads` lanes, 8 columns per lane, at most `cap` workgroups per job
static uint32_t job_wgs(const vlmc_select_job &j, int threads, int cap) {
    const int64_t chunks = j.out_features * ((j.in_features + 7) / 8);
    int64_t n = (chunks + threads - 1) / threads;
    return uint32_t(n < 1 ? 1 : (n > cap ? cap : n));
}

template <typename T>
static int launch_matrix(const vlmc_select_job *jobs, const int *idx, int n, int apply_zero, hipStream_t st) {
    SelBatch b;
    b.n = n; b.apply_zero = apply_zero;
    b.p0 = uint32_t(env_int("VLMC_MATRIX_FORCE_SLOW", 0)); b.p1 = 0;
    bool aligned = true;
    uint32_t wgs = 0;
    // persistent workgroups: 2 x 1024 lanes per CU over the whole launch, shared out by matrix size, so the
    // fixed per-workgroup latencies (control loads, LDS clear/flush, done-counter atomic) are paid once
    const int budget = env_int("VLMC_MATRIX_WGS", 512);
    int64_t all_chunks = 0;
    for (int i = 0; i < n; ++i) all_chunks += jobs[idx[i]].out_features * ((jobs[idx[i]].in_features + 7) / 8);
    for (int i = 0; i < n; ++i) {
        fill_job(b.job[i], jobs[idx[i]]);
        const int64_t chunks = jobs[idx[i]].out_features * ((jobs[idx[i]].in_features + 7) / 8);
        int64_t share = (chunks * budget + all_chunks - 1) / all_chunks;
        const uint32_t cap = job_wgs(jobs[idx[i]], 1024, kMatrixParts);
        b.job[i].nwg = uint32_t(share < 1 ? 1 : (share > cap ? cap : share));
        wgs += b.job[i].nwg;
        b.job[i].unit_end = wgs;
        aligned = aligned && job_aligned(jobs[idx[i]]);
    }
    int64_t max_in = 0;
    for (int i = 0; i < n; ++i) max_in = jobs[idx[i]].in_features > max_in ? jobs[idx[i]].in_features : max_in;
    if (env_int("VLMC_MATRIX_FUSED", 1) && max_in <= kFusedMaxIn) {
        // one workgroup per CU at most: co-resident by construction (see matrix_fused_kernel)
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus < n) {
            set_error("vlmc_wanda_select: cannot query the device's CU count");
            return VLMC_EHIP;
        }
        int fbudget = env_int("VLMC_MATRIX_FUSED_WGS", cus) < cus ? env_int("VLMC_MATRIX_FUSED_WGS", cus) : cus;
        if (fbudget > kMaxFusedWgs) fbudget = kMaxFusedWgs;
        if (fbudget < n) fbudget = n;
        uint32_t total = 0;
        int big = 0;
        for (int i = 0; i < n; ++i) {
            const int64_t chunks = jobs[idx[i]].out_features * ((jobs[idx[i]].in_features + 7) / 8);
            int64_t share = chunks * fbudget / all_chunks;
            const uint32_t cap = job_wgs(jobs[idx[i]], 1024, kMatrixParts);
            b.job[i].nwg = uint32_t(share < 1 ? 1 : (share > cap ? cap : share));
            total += b.job[i].nwg;
            if (b.job[i].nwg > b.job[big].nwg) big = i;
        }
        while (total > uint32_t(fbudget > n ? fbudget : n)) {               // the minimum of one per job overshot the budget
            big = 0;
            for (int i = 1; i < n; ++i) if (b.job[i].nwg > b.job[big].nwg) big = i;
            if (b.job[big].nwg <= 1) break;
            --b.job[big].nwg; --total;
        }
        wgs = 0;
        for (int i = 0; i < n; ++i) { wgs += b.job[i].nwg; b.job[i].unit_end = wgs; }
        // chunks a lane keeps in registers (128 VGPRs per lane at 1024 lanes per CU): 14 x 16 B covers a ViT-g block
        constexpr int R = sizeof(typename T::raw) == 2 ? 14 : 8;
        if (aligned) VLMC_LAUNCH_TIMED((matrix_fused_kernel<T, true, R>), dim3(wgs), dim3(1024), st, b);
        else VLMC_LAUNCH_TIMED((matrix_fused_kernel<T, false, 8>), dim3(wgs), dim3(1024), st, b);
        return VLMC_OK;
    }
    hipLaunchKernelGGL((matrix_sample_kernel<T>), dim3(unsigned(n)), dim3(1024), 0, st, b);
    if (aligned) {
        hipLaunchKernelGGL((matrix_count_kernel<T, true>), dim3(wgs), dim3(1024), 0, st, b);
        hipLaunchKernelGGL((matrix_apply_kernel<T, true>), dim3(wgs), dim3(1024), 0, st, b);
    } else {
        hipLaunchKernelGGL((matrix_count_kernel<T, false>), dim3(wgs), dim3(1024), 0, st, b);
        hipLaunchKernelGGL((matrix_apply_kernel<T, false>), dim3(wgs), dim3(1024), 0, st, b);
    }
    hipLaunchKernelGGL((matrix_resolve_kernel<T>), dim3(unsigned(n)), dim3(1024), 0, st, b);
    return VLMC_OK;
}

template <typename T>
static int launch_nm(const vlmc_select_job *jobs, const int *idx, int n, int prune_n, int prune_m, int apply_zero,
                     hipStream_t st) {
    SelBatch b;
    b.n = n; b.apply_zero = apply_zero;
    b.p0 = uint32_t(prune_n); b.p1 = 0;
    bool aligned = true;
    uint32_t wgs = 0;
    for (int i = 0; i < n; ++i) {
        fill_job(b.job[i], jobs[idx[i]]);
        b.job[i].nwg = job_wgs(jobs[idx[i]], 256, kNmParts);
        wgs += b.job[i].nwg;
        b.job[i].unit_end = wgs;
        aligned = aligned && job_aligned(jobs[idx[i]]);
    }
#define VLMC_NM(M)                                                                                         \
    do {                                                                                                   \
        if (aligned) hipLaunchKernelGGL((nm_kernel<T, true, M>), dim3(wgs), dim3(256), 0, st, b);          \
        else hipLaunchKernelGGL((nm_kernel<T, false, M>), dim3(wgs), dim3(256), 0, st, b);                 \
    } while (0)
    switch (prune_m) {
        case 2: VLMC_NM(2); break;
        case 4: VLMC_NM(4); break;
        case 8: VLMC_NM(8); break;
        default: set_error("vlmc_wanda_select: n:m with m=%d unsupported (m must be 2, 4 or 8)", prune_m); return VLMC_EINVAL;
    }
#undef VLMC_NM
    return VLMC_OK;
}

template <typename T>
static int select_typed(const vlmc_select_job *jobs, int n_jobs, int mode, int prune_n, int prune_m, int apply_zero,
                        hipStream_t st) {
    // launch groups: consecutive runs of <= kMaxSelJobs jobs; SEL_ROW additionally needs equal (in, k, alignment).
    // (Running the groups of one call side by side on a second stream was measured: the event fork/join costs
    // more than the overlap of T5's small `wo` group with the large one gains -- 46 vs 35 us per block.)
    if constexpr (sizeof(typename T::raw) == 2) {              // (fp32 rows need twice the registers: separate launches)
        if (mode == VLMC_SEL_ROW && mixed_rows_eligible(jobs, n_jobs)) {
            const int rc = launch_rows_mixed<T>(jobs, n_jobs, apply_zero, st);
            if (rc != VLMC_OK) return rc;
            VLMC_HIP_CHECK_LAUNCH("vlmc_wanda_select");
            return VLMC_OK;
        }
    }
    int idx[kMaxSelJobs];
    bool done[256] = {false};
    for (int s0 = 0; s0 < n_jobs; ++s0) {
        if (done[s0]) continue;
        int n = 0;
        for (int i = s0; i < n_jobs && n < kMaxSelJobs; ++i) {
            if (done[i]) continue;
            if (mode == VLMC_SEL_ROW && (jobs[i].in_features != jobs[s0].in_features || jobs[i].k != jobs[s0].k ||
                                         job_aligned(jobs[i]) != job_aligned(jobs[s0])))
                continue;
            idx[n++] = i;
            done[i] = true;
        }
        int rc;
        if (mode == VLMC_SEL_ROW) rc = launch_rows<T>(jobs, idx, n, apply_zero, st);
        else if (mode == VLMC_SEL_MATRIX) rc = launch_matrix<T>(jobs, idx, n, apply_zero, st);
        else rc = launch_nm<T>(jobs, idx, n, prune_n, prune_m, apply_zero, st);
        if (rc != VLMC_OK) return rc;
    }
    VLMC_HIP_CHECK_LAUNCH("vlmc_wanda_select");
    return VLMC_OK;
}

}  // namespace vlmc

using namespace vlmc;

extern "C" size_t vlmc_wanda_select_workspace(int mode, int64_t out_features, int64_t in_features) {
    if (out_features <= 0 || in_features <= 0 || mode < 0 || mode > 2) return 0;
    return ws_bytes(mode);
}

extern "C" int64_t vlmc_wanda_select_partials(int mode, int64_t out_features, int64_t in_features) {
    if (out_features <= 0 || in_features <= 0 || mode < 0 || mode > 2) return 0;
    return n_partials(mode, out_features);
}

extern "C" int vlmc_wanda_select_batch(const vlmc_select_job *jobs, int n_jobs, int dtype, int mode, int n, int m,
                                       int apply_zero, void *stream) {
    VLMC_REQUIRE(jobs && n_jobs > 0 && n_jobs <= 256, "vlmc_wanda_select_batch: 1..256 jobs expected (got %d)", n_jobs);
    VLMC_REQUIRE(mode >= 0 && mode <= 2, "vlmc_wanda_select: unknown mode %d", mode);
    for (int i = 0; i < n_jobs; ++i) {
        const vlmc_select_job &j = jobs[i];
        VLMC_REQUIRE(j.W && j.sqrt_scaler && j.mask, "vlmc_wanda_select: null pointer (job %d)", i);
        VLMC_REQUIRE(j.out_features > 0 && j.in_features > 0 && j.ldw >= j.in_features && j.ldw < (int64_t(1) << 32),
                     "vlmc_wanda_select: bad shape out=%lld in=%lld ldw=%lld", (long long)j.out_features,
                     (long long)j.in_features, (long long)j.ldw);
        VLMC_REQUIRE(j.out_features * j.in_features < (int64_t(1) << 32), "vlmc_wanda_select: more than 2^32 weights");
        if (mode == VLMC_SEL_ROW) {
            VLMC_REQUIRE(j.k >= 0 && j.k <= j.in_features, "vlmc_wanda_select: row k=%lld outside [0,%lld]", (long long)j.k,
                         (long long)j.in_features);
        } else if (mode == VLMC_SEL_MATRIX) {
            VLMC_REQUIRE(j.k >= 0 && j.k < j.out_features * j.in_features, "vlmc_wanda_select: matrix k=%lld outside [0,%lld)",
                         (long long)j.k, (long long)(j.out_features * j.in_features));
            const size_t need = ws_bytes(mode);
            VLMC_REQUIRE(j.workspace, "vlmc_wanda_select: null workspace");
            VLMC_REQUIRE((reinterpret_cast<uintptr_t>(j.workspace) % 256) == 0, "vlmc_wanda_select: workspace not 256-B aligned");
            if (j.workspace_bytes < need) {
                set_error("vlmc_wanda_select: workspace %zu B < required %zu B", j.workspace_bytes, need);
                return VLMC_EWORKSPACE;
            }
            for (int p = 0; p < i; ++p)
                VLMC_REQUIRE(jobs[p].workspace != j.workspace, "vlmc_wanda_select_batch: jobs %d and %d share a workspace", p, i);
        } else {
            VLMC_REQUIRE(m > 0 && n >= 0 && n <= m && j.in_features % m == 0,
                         "vlmc_wanda_select: bad n:m = %d:%d for in_features %lld", n, m, (long long)j.in_features);
        }
    }
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case VLMC_F32: return select_typed<f32_t>(jobs, n_jobs, mode, n, m, apply_zero, st);
        case VLMC_F16: return select_typed<f16_t>(jobs, n_jobs, mode, n, m, apply_zero, st);
        case VLMC_BF16: return select_typed<bf16_t>(jobs, n_jobs, mode, n, m, apply_zero, st);
    }
    set_error("vlmc_wanda_select: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}

extern "C" int vlmc_wanda_select(void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw,
                                 const float *sqrt_scaler, int mode, int64_t k, int n, int m, int apply_zero,
                                 uint8_t *mask, double *score_partials, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    const vlmc_select_job j{W, out_features, in_features, ldw, sqrt_scaler, k, mask, score_partials, workspace,
                            workspace_bytes};
    return vlmc_wanda_select_batch(&j, 1, dtype, mode, n, m, apply_zero, stream);
}
