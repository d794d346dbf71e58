// K12-K13, fast path: DSnoT refinement with the sorted lists cut to what the cycles can reach.
//
// The reference sorts every row three times and then walks the sorted lists for at most `max_cycle_time`
// (100) cycles; a cycle only ever takes the NEXT element of one of six lists (regrow: ascending / descending G;
// prune: ascending / descending wanda metric among the kept columns with negative resp. positive D).  So only
// the first max_cycle entries of each list matter.  One workgroup per row:
//   1. keys in registers (as in dsnot.hip)
//   2. per list: a counting sort of the list's end (dl_extract): the occupied key range is cut into 1024 equal bins,
//      the first bin at which the running count reaches max_cycle is the cut-off, the candidates at or below it are placed
//      by bin and ranked inside their bin.  Heavy ties / NaN keys (more than 192 candidates) take the exact radix route:
//      radix select (4 x 8 key bits) of the max_cycle-th smallest key, ties cut by column, gather, rank sort.
//      (Round 2: the radix route alone was ~21 k vector instructions per wave and row, profiles/r02_dsnot_roofline.md.)
//      n:m: each regrow entry carries its m-group's kept columns sorted by metric (the prune candidates).
//   3. wave 0 walks the cycles reading the lists: O(1) per cycle.
// Same events as dsnot_simulate_kernel (dsnot.hip), which remains the path for max_cycle > 128 and the
// cross-check in tests/test_dsnot_gpu.py; 30-100 x faster at model widths.
#pragma once
#include <cstdlib>

#include "common.hpp"
#include "topk_order.hpp"

namespace vlmc {

__device__ __forceinline__ uint32_t dl_signed_key(float x) {          // dsnot.hip: signed_key
    if (x != x) return 0xFFFFFFFFu;
    x = x + 0.f;
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ uint32_t dl_wave_incl_scan(uint32_t v) {
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, false));   // row_shr:1
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, false));   // row_shr:2
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, false));   // row_shr:4
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, false));   // row_shr:8
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31
    return v;
}

// The compiler otherwise keeps every per-element predicate (a 64-bit lane mask each) and every bin index of one pass alive
// for the next pass over the same registers: hundreds of spilled SGPRs and 2x the VGPRs (occupancy 1-2 waves per SIMD).  A
// pass starts from an opaque copy of its mask / range so that what it needs is recomputed (two instructions per element).
__device__ __forceinline__ uint32_t dl_opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t dl_uniform(uint32_t v) {          // a wave-uniform value, held in an SGPR
    v = uint32_t(__builtin_amdgcn_readfirstlane(int(v)));
    asm volatile("" : "+s"(v));
    return v;
}

constexpr int kListCap = 128;        // entries per list = max supported max_cycle
constexpr int kGroupMax = 8;         // m of n:m

struct ListEntry {
    uint32_t col;
    float d;
};
struct GroupInfo {                   // n:m: the kept columns of an entry's m-group in the order successive visits take them
    uint16_t col[kGroupMax];
    float d[kGroupMax];
    float dx;                        // D of the column a visit of the exhausted group picks (all metrics +inf)
    uint16_t colx;
    uint16_t n;
};

constexpr int kFastBinsLog2 = 10;
constexpr int kFastBins = 1 << kFastBinsLog2;  // fast route: one linear histogram over the occupied key range
constexpr int kRawCap = 192;                   // ... and up to this many candidates (the cut-off bin brings a few extra); 512 for wide rows, see ListSmem

template <int NW, bool NM> struct ListSmem {
    // Rows of more than 4096 columns (4+ waves) put several hundred keys into the half-octave bin that completes a list of 100
    // (the signed regrow keys span the whole key range: 1024 bins = half an octave each): 192 slots sent 11008-column rows to
    // the radix route for most lists (4096 x 11008: 1384 -> 1100 us with 512); narrow rows keep 192 (LDS is their occupancy).
    static constexpr int RAWCAP = NW >= 4 ? 512 : kRawCap;
    uint32_t hist[kFastBins + 1 + 64];
    uint32_t red[24];
    uint32_t rawk[2][RAWCAP], rawc[2][RAWCAP];       // [1]: the descending list of dl_extract_both
    ListEntry list[NM ? 2 : 6][kListCap];
    uint32_t wmin[NW], wmax[NW];
    GroupInfo grp[NM ? 2 : 1][NM ? kListCap : 1];
    uint32_t cyc_group[NM ? kListCap : 1];
    float fsum[NW];
    uint32_t cnt3[3][NW];
    uint32_t nlist[6];
};

template <int NW> __device__ __forceinline__ void dl_sync() {
    if constexpr (NW > 1) __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// rank-r (0-based) smallest value of f(i) over the elements with mask bit i; also the count of values below it.
// f values are < 2^(8*PASSES).  All threads of the workgroup call it with the same r.
template <int E, int NT, int NW, int PASSES, typename F, typename S, typename R>
__device__ __forceinline__ uint32_t dl_radix_kth(F f, uint32_t mask, uint32_t r, S &sm, uint32_t &below, R new_pass) {
    const int tid = threadIdx.x;
    uint32_t prefix = 0, pmask = 0;
    below = 0;
    for (int sh = 8 * (PASSES - 1); sh >= 0; sh -= 8) {
        for (int i = tid; i < 256; i += NT) sm.hist[i] = 0;
        dl_sync<NW>();
        const uint32_t mk = dl_opaque(mask);
        new_pass();                                                  // (nothing of f is hoisted out of the loop over the passes)
#pragma unroll
        for (int i = 0; i < E; ++i)
            if ((mk >> i) & 1u) {
                const uint32_t v = f(i);
                if ((v & pmask) == prefix) atomicAdd(&sm.hist[(v >> sh) & 255u], 1u);
            }
        dl_sync<NW>();
        if (tid < 64) {                                              // wave 0: 4 bins per lane
            uint32_t h[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = sm.hist[tid * 4 + i];
            const uint32_t s = h[0] + h[1] + h[2] + h[3];
            const uint32_t incl = dl_wave_incl_scan(s);
            const uint32_t lo = incl - s;
            if (lo <= r && r < incl) {
                uint32_t cum = lo;
                int i = 0;
                for (; i < 3; ++i) {
                    if (cum + h[i] > r) break;
                    cum += h[i];
                }
                sm.red[0] = uint32_t(tid * 4 + i);
                sm.red[1] = cum;
            }
        }
        dl_sync<NW>();
        const uint32_t bin = sm.red[0], before = sm.red[1];
        prefix |= bin << sh;
        pmask |= 0xFFu << sh;
        below += before;
        r -= before;
        dl_sync<NW>();
    }
    return prefix;
}

template <int NT, int NW, typename S> __device__ __forceinline__ uint32_t dl_block_sum(uint32_t v, S &sm, int slot) {
    v = wave_sum_u32(v);
    if constexpr (NW > 1) {
        if ((threadIdx.x & 63) == 0) sm.cnt3[slot][threadIdx.x >> 6] = v;
        __syncthreads();
        v = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += sm.cnt3[slot][w];
        __syncthreads();
    }
    return v;
}

// wave-wide min / max of a 32-bit unsigned (result valid in every lane)
__device__ __forceinline__ uint32_t dl_wave_min(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = uint32_t(__shfl_xor(int(v), off, 64));
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t dl_wave_max(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = uint32_t(__shfl_xor(int(v), off, 64));
        v = o > v ? o : v;
    }
    return v;
}

// The K smallest (want_max: largest) (key, column) pairs among the masked elements -> out[0..n) in that order.
// The mask, the direction and K are RUN-TIME values, so that the kernel holds one copy of this code per key vector and
// loops over that vector's lists (seven inlined copies were 287 KB of code per kernel, 4.5 x the instruction cache).
// Returns n = min(K, number of masked elements).  `dval(col)` gives the entry's D = w * mean (evaluated for the <= K placed
// entries, from the row in memory: an array of D in registers cost a wave of occupancy); `place(col, rank)` is called by
// one thread for each list entry (n:m: it builds the entry's group payload).
//
// Fast route (fast != 0): a counting sort.  The occupied key range [lo, hi] of the masked elements is cut into kFastBins
// equal bins; wave 0 turns the histogram into bin start offsets and finds the cut-off bin (the first one at which the
// running count reaches n); every element at or below it takes a slot inside its bin's range (LDS atomic), which orders
// the candidates across bins, and each candidate's exact rank is its bin's start + its rank by (key, column) among the
// few entries of its own bin.  Ranks >= n are dropped.  More than kRawCap candidates (heavy ties: zero weights, dead
// channels; NaN keys stretching the range) -> the radix route below, which is exact for any input.
template <int E, int NT, int NW, typename S, typename DVal, typename Place>
__device__ __forceinline__ uint32_t dl_extract(const uint32_t (&key)[E], DVal dval, uint32_t mask, uint32_t want_max, uint32_t K, S &sm,
                                               ListEntry *out, Place place, int fast) {
    const int tid = threadIdx.x;
    const uint32_t avail = dl_block_sum<NT, NW>(uint32_t(__popc(mask)), sm, 0);
    const uint32_t n = avail < K ? avail : K;
    if (n == 0) return 0;
    uint32_t flip = want_max ? 0xFFFFFFFFu : 0u;
    uint32_t tid8 = uint32_t(tid) * 8u;
    auto colof = [&](int i) { return uint32_t((i / 8) * NT * 8 + (i % 8)) + tid8; };
    auto kf = [&](int i) { return key[i] ^ flip; };
    auto cf = [&](int i) { return (colof(i) ^ flip) & 0x3FFFu; };
    // a pass starts from opaque copies of the direction and the lane's column base: no flipped key or column number of an
    // earlier pass (or, hoisted out of the loop over the lists, of all of them) is kept in a register
    auto new_pass = [&]() { flip = dl_uniform(flip); tid8 = dl_opaque(tid8); };
    if (fast) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        const uint32_t mk0 = dl_opaque(mask);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t kk = kf(i);
            const bool in = (mk0 >> i) & 1u;
            lo = (in && kk < lo) ? kk : lo;
            hi = (in && kk > hi) ? kk : hi;
        }
        lo = dl_wave_min(lo);
        hi = dl_wave_max(hi);
        if constexpr (NW > 1) {
            if ((tid & 63) == 0) { sm.wmin[tid >> 6] = lo; sm.wmax[tid >> 6] = hi; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t a = sm.wmin[w], b = sm.wmax[w];
                lo = a < lo ? a : lo;
                hi = b > hi ? b : hi;
            }
        }
        lo = dl_uniform(lo);
        const uint32_t span = dl_uniform(hi) - lo;                      // (span >> shift) < kFastBins
        uint32_t shift = span < uint32_t(kFastBins) ? 0u : uint32_t(32 - __builtin_clz(span)) - uint32_t(kFastBinsLog2);
        for (int i = tid; i < kFastBins; i += NT) sm.hist[i] = 0;
        dl_sync<NW>();
        const uint32_t mk1 = dl_opaque(mask);
        new_pass();
#pragma unroll
        for (int i = 0; i < E; ++i) {                                   // branch-free: the others count into a spare bin of their lane
            const uint32_t bin = (kf(i) - lo) >> shift;
            atomicAdd(&sm.hist[((mk1 >> i) & 1u) ? bin : uint32_t(kFastBins + 1 + (tid & 63))], 1u);
        }
        dl_sync<NW>();
        lo = dl_uniform(lo);                                            // (the bins are recomputed in the gather, not kept)
        shift = dl_uniform(shift);
        if (tid < 64) {                                                 // wave 0: kFastBins / 64 bins per lane
            constexpr int PER = kFastBins / 64;
            uint32_t h[PER];
            uint32_t ssum = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) { h[i] = sm.hist[tid * PER + i]; ssum += h[i]; }
            const uint32_t incl = dl_wave_incl_scan(ssum);
            uint32_t cum = incl - ssum;
            const bool mine = cum < n && n <= incl;                     // this lane's bins bring the count to n
            bool found = false;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                cum += h[i];
                sm.hist[tid * PER + i] = cum;                           // inclusive prefix: bin b's range is [hist[b-1] - cnt_b .. )
                if (mine && !found && cum >= n) {
                    found = true;
                    sm.red[3] = uint32_t(tid * PER + i);                // the cut-off bin
                    sm.red[4] = cum;                                    // candidates
                }
            }
        }
        dl_sync<NW>();
        const uint32_t cut = sm.red[3], m = sm.red[4];
        if (m <= uint32_t(S::RAWCAP)) {
            // slots are handed out from the END of each bin's range downwards: hist[b] (inclusive prefix) counts down to
            // the bin's start, so afterwards start(b) = hist[b] and end(b) = start(b + 1) = hist[b + 1] (or m at the cut).
            const uint32_t mk2 = dl_opaque(mask);
            new_pass();
#pragma unroll
            for (int i = 0; i < E; ++i)
                if ((mk2 >> i) & 1u) {
                    const uint32_t kk = kf(i);
                    const uint32_t bin = (kk - lo) >> shift;
                    if (bin <= cut) {
                        const uint32_t pos = atomicSub(&sm.hist[bin], 1u) - 1u;
                        sm.rawk[0][pos] = kk;
                        sm.rawc[0][pos] = cf(i);
                    }
                }
            dl_sync<NW>();
            for (uint32_t p = tid; p < m; p += NT) {
                const uint32_t kp = sm.rawk[0][p], cp = sm.rawc[0][p];
                const uint32_t bin = (kp - lo) >> shift;
                const uint32_t s0 = sm.hist[bin], e0 = bin == cut ? m : sm.hist[bin + 1];
                uint32_t rank = s0;
                for (uint32_t q = s0; q < e0; ++q) {
                    const uint32_t kq = sm.rawk[0][q], cq = sm.rawc[0][q];
                    rank += (kq < kp || (kq == kp && cq < cp)) ? 1u : 0u;
                }
                if (rank < n) {
                    const uint32_t col = (cp ^ flip) & 0x3FFFu;
                    out[rank] = ListEntry{col, dval(col)};
                    place(col, rank);
                }
            }
            dl_sync<NW>();
            return n;
        }
        dl_sync<NW>();
    }
#ifdef DL_EXP_NORADIX
    return n;
#endif
    // ---- exact route for any input: radix select of the n-th key, ties cut by column --------------------------------------
    uint32_t c_lt;
    const uint32_t X = dl_radix_kth<E, NT, NW, 4>(kf, mask, n - 1, sm, c_lt, new_pass);
    // ties with X: all of them, or the (n - c_lt) first by (flipped) column
    uint32_t tie = 0;
    const uint32_t mk3 = dl_opaque(mask);
    new_pass();
#pragma unroll
    for (int i = 0; i < E; ++i)
        if (((mk3 >> i) & 1u) && kf(i) == X) tie |= 1u << i;
    const uint32_t n_tie = dl_block_sum<NT, NW>(uint32_t(__popc(tie)), sm, 1);
    uint32_t Y = 0x3FFFu;
    if (n_tie > n - c_lt) {
        uint32_t dummy;
        Y = dl_radix_kth<E, NT, NW, 2>(cf, tie, n - c_lt - 1, sm, dummy, new_pass);
    }
    if (tid == 0) sm.red[2] = 0;
    dl_sync<NW>();
    const uint32_t mk4 = dl_opaque(mask);
    new_pass();
#pragma unroll
    for (int i = 0; i < E; ++i)
        if ((mk4 >> i) & 1u) {
            const uint32_t kk = kf(i);
            if (kk < X || (kk == X && cf(i) <= Y)) {
                const uint32_t pos = atomicAdd(&sm.red[2], 1u);
                sm.rawk[0][pos] = kk;
                sm.rawc[0][pos] = cf(i);
            }
        }
    dl_sync<NW>();
    // rank sort of the n <= 128 gathered entries.  One wave per row: lane l holds entries l and l + 64 in registers and
    // the entries are broadcast one by one through v_readlane (no LDS latency in the loop: 757 -> 596 us per 6144x1408
    // linear); several waves: every lane ranks one entry against the LDS copy (faster than serialising on wave 0).
    if constexpr (NW > 1) {
        for (uint32_t p = tid; p < n; p += NT) {
            const uint32_t kp = sm.rawk[0][p], cp = sm.rawc[0][p];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < n; ++q) {
                const uint32_t kq = sm.rawk[0][q], cq = sm.rawc[0][q];
                rank += (kq < kp || (kq == kp && cq < cp)) ? 1u : 0u;
            }
            const uint32_t col = (cp ^ flip) & 0x3FFFu;
            out[rank] = ListEntry{col, dval(col)};
            place(col, rank);
        }
    } else if (tid < 64) {
        const uint32_t p0 = uint32_t(tid), p1 = uint32_t(tid) + 64u;
        const uint32_t k0 = p0 < n ? sm.rawk[0][p0] : 0xFFFFFFFFu, c0 = p0 < n ? sm.rawc[0][p0] : 0xFFFFFFFFu;
        const uint32_t k1 = p1 < n ? sm.rawk[0][p1] : 0xFFFFFFFFu, c1 = p1 < n ? sm.rawc[0][p1] : 0xFFFFFFFFu;
        uint32_t rank0 = 0, rank1 = 0;
        const uint32_t n_lo = n < 64u ? n : 64u;
        for (uint32_t q = 0; q < n_lo; ++q) {
            const uint32_t kq = uint32_t(__builtin_amdgcn_readlane(int(k0), int(q))), cq = uint32_t(__builtin_amdgcn_readlane(int(c0), int(q)));
            rank0 += (kq < k0 || (kq == k0 && cq < c0)) ? 1u : 0u;
            rank1 += (kq < k1 || (kq == k1 && cq < c1)) ? 1u : 0u;
        }
        for (uint32_t q = 64; q < n; ++q) {
            const uint32_t kq = uint32_t(__builtin_amdgcn_readlane(int(k1), int(q - 64u))), cq = uint32_t(__builtin_amdgcn_readlane(int(c1), int(q - 64u)));
            rank0 += (kq < k0 || (kq == k0 && cq < c0)) ? 1u : 0u;
            rank1 += (kq < k1 || (kq == k1 && cq < c1)) ? 1u : 0u;
        }
        if (p0 < n) { const uint32_t col = (c0 ^ flip) & 0x3FFFu; out[rank0] = ListEntry{col, dval(col)}; place(col, rank0); }
        if (p1 < n) { const uint32_t col = (c1 ^ flip) & 0x3FFFu; out[rank1] = ListEntry{col, dval(col)}; place(col, rank1); }
    }
    dl_sync<NW>();
    return n;
}

// BOTH ends of one key vector's order among the masked elements in one go: outA[0..n) = the n smallest (key, column)
// pairs ascending, outD[0..n) = the n largest, descending in key and, among equal keys, in column -- what dl_extract gives
// for want_max = 0 and 1 -- from ONE min / max reduction, ONE histogram pass, one scan and one gather pass (the two lists of
// a key vector were two full extractions: seven per row, now three of these and a min-reduction for K0).  The bins are cut
// from the unflipped keys; the ascending list takes the bins up to the one at which the running count reaches n, the
// descending list the bins from the one that holds ascending position avail - n upwards; an element's exact rank is its
// bin's offset + its rank among the few entries of its own bin.  Falls back to two dl_extract calls when the two ends meet
// (few masked elements) or either end brings more than kRawCap candidates (ties, NaN / Inf keys).
// `place(which, col, rank)`: called by one thread per entry; which = 0 the ascending list, 1 the descending one.
template <int E, int NT, int NW, typename S, typename DVal, typename Place>
__device__ __forceinline__ uint32_t dl_extract_both(const uint32_t (&key)[E], DVal dval, uint32_t mask, uint32_t K, S &sm, ListEntry *outA,
                                                    ListEntry *outD, Place place, int fast, uint32_t flood_key, uint32_t wlog) {
    // `wlog` != 0: WINDOWED bins.  Signed keys (the regrow vector G) span the whole 32-bit range -- 1024 linear bins are a quarter
    // of an octave each, the 100 largest of a few thousand values share two or three of them (hundreds of candidates, LDS
    // atomics on a handful of addresses).  Both lists of G end at LARGE magnitudes, so only the 2^wlog keys next to the
    // minimum (512 bins) and next to the maximum (512 bins) are binned, 1/128 octave each for wlog = 25; everything between
    // is one uncounted gap.  An end whose window holds fewer than K elements falls back to the single-list code.
    const int tid = threadIdx.x;
    const uint32_t avail = dl_block_sum<NT, NW>(uint32_t(__popc(mask)), sm, 0);
    const uint32_t n = avail < K ? avail : K;
    if (n == 0) return 0;
    if (fast && avail >= 2u * n) {
        constexpr uint32_t kNoBin = 0xFFFFFFFFu;
        uint32_t tid8 = uint32_t(tid) * 8u;
        auto colof = [&](int i) { return uint32_t((i / 8) * NT * 8 + (i % 8)) + tid8; };
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        const uint32_t mk0 = dl_opaque(mask);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t kk = key[i];
            const bool in = (mk0 >> i) & 1u;
            lo = (in && kk < lo) ? kk : lo;
            hi = (in && kk > hi) ? kk : hi;
        }
        lo = dl_wave_min(lo);
        hi = dl_wave_max(hi);
        if constexpr (NW > 1) {
            if ((tid & 63) == 0) { sm.wmin[tid >> 6] = lo; sm.wmax[tid >> 6] = hi; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t a = sm.wmin[w], b = sm.wmax[w];
                lo = a < lo ? a : lo;
                hi = b > hi ? b : hi;
            }
        }
        lo = dl_uniform(lo);
        hi = dl_uniform(hi);
        const uint32_t span = hi - lo;
        uint32_t shift = span < uint32_t(kFastBins) ? 0u : uint32_t(32 - __builtin_clz(span)) - uint32_t(kFastBinsLog2);
        const bool win = wlog != 0 && (span >> 1) >= (1u << wlog);        // the two windows do not meet
        const uint32_t W = 1u << wlog, wsh = wlog - 9u;
        auto binof = [&](uint32_t k) -> uint32_t {
            if (!win) return (k - lo) >> shift;
            const uint32_t da = k - lo, dd = hi - k;
            return da < W ? da >> wsh : (dd < W ? uint32_t(kFastBins - 1) - (dd >> wsh) : kNoBin);
        };
        for (int i = tid; i < kFastBins; i += NT) sm.hist[i] = 0;
        if (tid == 0) { sm.red[3] = kNoBin; sm.red[5] = kNoBin; }
        dl_sync<NW>();
        // `flood_key`: the ONE key half of the elements may share (the regrow keys of all kept columns are G = 0; zero weights have
        // metric 0): counted in a register and added to its bin once per thread.
        const uint32_t mk1 = dl_opaque(mask);
        uint32_t flood = 0;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t bin = binof(key[i]);
            const bool in = (mk1 >> i) & 1u, fl = in && key[i] == flood_key;
            flood += fl ? 1u : 0u;
            atomicAdd(&sm.hist[(in && !fl && bin != kNoBin) ? bin : uint32_t(kFastBins + 1 + (tid & 63))], 1u);
        }
        if (flood && binof(flood_key) != kNoBin) atomicAdd(&sm.hist[binof(flood_key)], flood);
        dl_sync<NW>();
        lo = dl_uniform(lo);
        hi = dl_uniform(hi);
        shift = dl_uniform(shift);
        const uint32_t need_d = avail - n;                              // ascending position of the descending list's last entry
        if (tid < 64) {
            constexpr int PER = kFastBins / 64;
            uint32_t h[PER];
            uint32_t ssum = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) { h[i] = sm.hist[tid * PER + i]; ssum += h[i]; }
            const uint32_t incl0 = dl_wave_incl_scan(ssum);
            const uint32_t counted = uint32_t(__builtin_amdgcn_readlane(int(incl0), 63));
            // windowed: the uncounted gap sits between bins 511 and 512 (lanes 31 and 32) of the ascending order
            const uint32_t gap = (win && tid >= 32) ? avail - counted : 0u;
            const uint32_t incl = incl0 + gap;
            uint32_t cum = incl - ssum;
            const bool mine_a = cum < n && n <= incl;
            const bool mine_d = cum <= need_d && need_d < incl;
            bool found_a = false, found_d = false;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const uint32_t excl = cum;
                cum += h[i];
                sm.hist[tid * PER + i] = cum;                           // inclusive prefix
                if (mine_a && !found_a && cum >= n) {
                    found_a = true;
                    sm.red[3] = uint32_t(tid * PER + i);                // the ascending list's cut-off bin
                    sm.red[4] = cum;                                    // ... and its candidates
                }
                if (mine_d && !found_d && need_d < cum) {
                    found_d = true;
                    sm.red[5] = uint32_t(tid * PER + i);                // the descending list's lowest bin
                    sm.red[6] = avail - excl;                           // ... and its candidates
                }
            }
        }
        dl_sync<NW>();
        const uint32_t cut_a = sm.red[3], m_a = sm.red[4], cut_d = sm.red[5], m_d = sm.red[6];
        const bool ends_ok = cut_a != kNoBin && cut_d != kNoBin && cut_a < cut_d &&
                             (!win || (cut_a < uint32_t(kFastBins / 2) && cut_d >= uint32_t(kFastBins / 2)));
        if (ends_ok && m_a <= uint32_t(S::RAWCAP) && m_d <= uint32_t(S::RAWCAP)) {
            // slots are handed out from the END of each bin's range downwards (hist[b] counts down from the inclusive prefix to the
            // bin's start): ascending position p of a candidate; the descending list stores it at avail - 1 - p
            const uint32_t mk2 = dl_opaque(mask);
            tid8 = dl_opaque(tid8);
#pragma unroll
            for (int i = 0; i < E; ++i)
                if ((mk2 >> i) & 1u) {
                    const uint32_t kk = key[i];
                    const uint32_t bin = binof(kk);
                    if (bin != kNoBin && (bin <= cut_a || bin >= cut_d)) {
                        const uint32_t p = atomicSub(&sm.hist[bin], 1u) - 1u;
                        const uint32_t which = bin <= cut_a ? 0u : 1u;
                        const uint32_t pos = which ? avail - 1u - p : p;
                        sm.rawk[which][pos] = kk;
                        sm.rawc[which][pos] = colof(i);
                    }
                }
            dl_sync<NW>();
            for (uint32_t p = tid; p < m_a + m_d; p += NT) {
                const bool desc = p >= m_a;
                const uint32_t q0 = desc ? p - m_a : p;
                const uint32_t kp = sm.rawk[desc ? 1 : 0][q0], cp = sm.rawc[desc ? 1 : 0][q0];
                const uint32_t bin = binof(kp);
                const uint32_t excl = sm.hist[bin];                     // (counted down to the bin's start)
                uint32_t rank;
                if (!desc) {
                    const uint32_t e0 = bin == cut_a ? m_a : sm.hist[bin + 1];
                    rank = excl;
                    for (uint32_t q = excl; q < e0; ++q) {
                        const uint32_t kq = sm.rawk[0][q], cq = sm.rawc[0][q];
                        rank += (kq < kp || (kq == kp && cq < cp)) ? 1u : 0u;
                    }
                } else {
                    const uint32_t incl = bin == uint32_t(kFastBins - 1) ? avail : sm.hist[bin + 1];
                    rank = avail - incl;
                    for (uint32_t q = avail - incl; q < avail - excl; ++q) {
                        const uint32_t kq = sm.rawk[1][q], cq = sm.rawc[1][q];
                        rank += (kq > kp || (kq == kp && cq > cp)) ? 1u : 0u;
                    }
                }
                if (rank < n) {                                         // (one copy of dval / place for both lists: lanes of a wave hold both kinds)
                    (desc ? outD : outA)[rank] = ListEntry{cp, dval(cp)};
                    place(desc ? 1u : 0u, cp, rank);
                }
            }
            dl_sync<NW>();
            return n;
        }
        dl_sync<NW>();
    }
    auto place_a = [&](uint32_t col, uint32_t rank) { place(0u, col, rank); };
    auto place_d = [&](uint32_t col, uint32_t rank) { place(1u, col, rank); };
    dl_extract<E, NT, NW>(key, dval, mask, 0u, K, sm, outA, place_a, fast);
    return dl_extract<E, NT, NW>(key, dval, mask, 1u, K, sm, outD, place_d, fast);
}

__device__ __attribute__((noinline)) float dl_powf(float v, float p) { return powf(v, p); }   // (32 inlined copies otherwise)

template <typename T, int CH, int NW, bool NM>
// (one wave per row -- rows of <= 1024 columns -- and the two-wave n:m form end at 3 waves per SIMD at their register footprint; the
// wider forms reach 4: the attribute states what each instantiation achieves)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu((NW == 1 || (NM && NW == 2)) ? 3 : 4))) void dsnot_lists_kernel(
    const typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f, int64_t ldw, const uint8_t *__restrict__ keep0,
    const float *__restrict__ sqrt_scaler, const float *__restrict__ sum_row, const float *__restrict__ var_row, int use_wanda_init,
    int prune_m, int max_cycle, float thr, float pow_var, int without_same_sign, uint32_t *__restrict__ events,
    int32_t *__restrict__ t_row, int fast) {
    constexpr int NT = 64 * NW, E = CH * 8;
    __shared__ ListSmem<NW, NM> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int64_t nchunks = in_f / 8;
    // ---- 1. keys (identical to dsnot_simulate_kernel) -------------------------------------------------------------
    // One key vector lives in registers at a time: G (regrow) here; the wanda metric is computed from the row again
    // (L2) after the regrow lists are cut.  D = w * mean is needed per element only for its sign (two bit masks).
    const typename T::raw *wrow = W + row * ldw;
    const uint8_t *krow = keep0 + row * in_f;
    uint32_t live = 0, pruned0 = 0, negm = 0, posm = 0;
    float part = 0.f;
    uint32_t gk[E];
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        const int64_t c = int64_t(s) * NT + tid;
        if (c < nchunks) {
            const int64_t col0 = c * 8;
            Chunk8<T> raw = load_chunk8<T>(wrow + col0);
            const uint2 mm = *reinterpret_cast<const uint2 *>(krow + col0);
            uint8_t m[8];
            __builtin_memcpy(m, &mm, 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = s * 8 + j;
                const float w = to_f32<T>(raw.v[j]);
                const float d = ieee_mul(w, sum_row[col0 + j]);
                const bool pr = m[j] == 0;
                live |= 1u << i;
                if (pr) pruned0 |= 1u << i;
                else if (d < 0.f) negm |= 1u << i;
                else if (d > 0.f) posm |= 1u << i;
                float g = pr ? d : 0.f;
                if (pr) part = ieee_add(part, d);
                if (pow_var != 0.f) {
                    const float v = var_row[col0 + j];
                    g = ieee_div(g, pow_var == 1.f ? v : dl_powf(v, pow_var));
                }
                gk[i] = dl_signed_key(g);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) gk[s * 8 + j] = 0;
        }
    }
    auto dval = [&](uint32_t col) { return ieee_mul(to_f32<T>(wrow[col]), sum_row[col]); };
    float err;
    {
        float v = part;
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
        err = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)) +
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
        if constexpr (NW > 1) {
            if (lane == 0) sm.fsum[wave] = err;
            __syncthreads();
            err = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) err += sm.fsum[w];
            __syncthreads();
        }
    }
    const float sign0 = err > 0.f ? 1.f : (err < 0.f ? -1.f : 0.f);
    const uint32_t kept0 = live & ~pruned0;
    const uint32_t K = uint32_t(max_cycle);

    // ---- 2. the list heads ---------------------------------------------------------------------------------------
    uint32_t NP = 0, PP = 0, Z = 0, k0col = 0;
    float k0d = 0.f;
    if constexpr (NM) {
        // payload of a regrow entry: its m-group's kept columns in ascending (metric, column) order, D of its first column.
        // Built by the thread that places the entry, from the row in memory (same arithmetic as step 1; 2 x max_cycle
        // groups per row, where the per-element version inlined at every gather site was most of the kernel's code).
        auto payload = [&](uint32_t li, uint32_t col, uint32_t rank) {
            GroupInfo gi;
            const uint32_t g0 = col - col % uint32_t(prune_m);
            uint32_t key[kGroupMax];
            float d[kGroupMax];
            bool kept[kGroupMax];
#pragma unroll
            for (int a = 0; a < kGroupMax; ++a) {
                gi.col[a] = 0; gi.d[a] = 0.f;
                const bool in = a < prune_m;
                const uint32_t c = in ? g0 + uint32_t(a) : g0;
                const float w = to_f32<T>(wrow[c]);
                d[a] = ieee_mul(w, sum_row[c]);
                key[a] = score_key(use_wanda_init ? ieee_mul(fabsf(w), sqrt_scaler[c]) : fabsf(w));
                kept[a] = in && krow[c] != 0;
            }
            // visit v takes what `torch.topk(block, 1, largest=False)` returns for the group's current metrics (:517-519) -- kept
            // columns their metric, pruned and taken ones +inf; equal minima in the reference's CPU order (topk_order.hpp) --
            // and marks it taken (:531).  Once nothing kept is left every visit picks the same column of the all-+inf group.
            uint32_t v[kGroupMax];
            gi.n = 0;
#pragma unroll
            for (int a = 0; a < kGroupMax; ++a) {
                v[a] = kept[a] ? key[a] : 0x7F800000u;
                gi.n += kept[a] ? 1 : 0;
            }
#pragma unroll 1
            for (int visit = 0; visit <= kGroupMax; ++visit) {
                const int pick = torch_cpu_argmin(v, prune_m);
                float pd = 0.f;
#pragma unroll
                for (int a = 0; a < kGroupMax; ++a) pd = a == pick ? d[a] : pd;
                if (visit < int(gi.n)) {
#pragma unroll
                    for (int slot = 0; slot < kGroupMax; ++slot)
                        if (slot == visit) { gi.col[slot] = uint16_t(g0 + uint32_t(pick)); gi.d[slot] = pd; }
#pragma unroll
                    for (int a = 0; a < kGroupMax; ++a) v[a] = a == pick ? 0x7F800000u : v[a];
                } else {
                    gi.colx = uint16_t(g0 + uint32_t(pick));
                    gi.dx = pd;
                    break;
                }
            }
            sm.grp[li][rank] = gi;
        };
        {
            auto place = [&](uint32_t which, uint32_t col, uint32_t rank) { payload(which, col, rank); };
            const uint32_t got = dl_extract_both<E, NT, NW>(gk, dval, live, K, sm, sm.list[0], sm.list[1], place, fast, 0x80000000u, 25u);
            if (tid == 0) { sm.nlist[0] = got; sm.nlist[1] = got; }
        }
    } else {
        NP = dl_block_sum<NT, NW>(uint32_t(__popc(negm)), sm, 0);
        PP = dl_block_sum<NT, NW>(uint32_t(__popc(posm)), sm, 1);
        Z = dl_block_sum<NT, NW>(uint32_t(__popc(kept0)), sm, 2) - NP - PP;
        auto no_place = [](uint32_t, uint32_t, uint32_t) {};
        // lists 0/1: regrow candidates by G, ascending / descending; 2/3: kept columns with D < 0 by metric; 4/5: with D > 0;
        // 6 (K0): the kept column with the smallest wanda metric (head of the ascending list over ALL kept columns)
        {                                                              // (gk is dead after this: 32 registers less)
            const uint32_t got = dl_extract_both<E, NT, NW>(gk, dval, live, K, sm, sm.list[0], sm.list[1], no_place, fast, 0x80000000u, 25u);
            if (tid == 0) { sm.nlist[0] = got; sm.nlist[1] = got; }
        }
        // the prune lists' key: the wanda metric |w| * sqrt(scaler), from the row again
        uint32_t wk[E];
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int64_t c = int64_t(s) * NT + tid;
            if (c < nchunks) {
                const int64_t col0 = c * 8;
                Chunk8<T> raw = load_chunk8<T>(wrow + col0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    wk[s * 8 + j] = score_key(ieee_mul(fabsf(to_f32<T>(raw.v[j])), sqrt_scaler[col0 + j]));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wk[s * 8 + j] = 0;
            }
        }
#pragma unroll 1
        for (uint32_t li = 2; li < 6; li += 2) {                       // both ends of the negative-D pool, then of the positive-D pool
            const uint32_t msk = li < 4 ? negm : posm;
            const uint32_t got = dl_extract_both<E, NT, NW>(wk, dval, msk, K, sm, sm.list[li], sm.list[li + 1], no_place, fast, 0u, 0u);
            if (tid == 0) { sm.nlist[li] = got; sm.nlist[li + 1] = got; }
        }
        // K0, the kept column with the smallest (metric, column): a min-reduction of key : column pairs
        {
            uint32_t bk = 0xFFFFFFFFu, bc = 0xFFFFFFFFu;
            const uint32_t tid8 = uint32_t(tid) * 8u;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const uint32_t c = uint32_t((i / 8) * NT * 8 + (i % 8)) + tid8;
                const bool in = (kept0 >> i) & 1u;
                const bool better = in && (wk[i] < bk || (wk[i] == bk && c < bc));
                bk = better ? wk[i] : bk;
                bc = better ? c : bc;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t ok = uint32_t(__shfl_xor(int(bk), off, 64)), oc = uint32_t(__shfl_xor(int(bc), off, 64));
                const bool better = ok < bk || (ok == bk && oc < bc);
                bk = better ? ok : bk;
                bc = better ? oc : bc;
            }
            if constexpr (NW > 1) {
                __syncthreads();
                if (lane == 0) { sm.wmin[wave] = bk; sm.wmax[wave] = bc; }
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const uint32_t ok = sm.wmin[w], oc = sm.wmax[w];
                    const bool better = ok < bk || (ok == bk && oc < bc);
                    bk = better ? ok : bk;
                    bc = better ? oc : bc;
                }
            }
            k0col = bc;                                                // 0xFFFFFFFF: no kept column
            k0d = bc != 0xFFFFFFFFu ? dval(bc) : 0.f;
        }
    }
    dl_sync<NW>();
    if (wave != 0) return;
#ifdef DL_EXP_NOWALK
    if (max_cycle > 0) { if (tid == 0) t_row[row] = 1; return; }
#endif

    // ---- 3. the cycles, by wave 0 (uniform control flow) ------------------------------------------------------------
    bool u = true;
    int stop = 0x7FFFFFFF;
    uint32_t hpos = 0, tpos = 0, r_lo = 0, r_hi = 0, a_nlo = 0, a_nhi = 0, a_plo = 0, a_phi = 0;
    for (int t = 0; t < max_cycle; ++t) {
        const bool r_tail = err > 0.f;
        // head and tail walkers share one visited set: an entry is available while lo + hi < number of columns
        uint32_t rcol = 0xFFFFFFFFu;
        float rd = 0.f;
        uint32_t ridx = 0;
        {
            const uint32_t idx = r_tail ? r_hi : r_lo;
            if (idx < sm.nlist[r_tail ? 1 : 0] && r_lo + r_hi < uint32_t(in_f)) {
                const ListEntry e = sm.list[r_tail ? 1 : 0][idx];
                rcol = e.col; rd = e.d; ridx = idx;
            }
            if (r_tail) ++r_hi; else ++r_lo;
        }
        uint32_t pcol = 0xFFFFFFFFu;
        float pd = 0.f;
        if constexpr (NM) {
            const uint32_t g0 = rcol - rcol % uint32_t(prune_m);
            // visits of this group so far = entries of its sorted kept list already taken
            // (both halves of the history are requested at once -- max_cycle <= 128 = 2 x 64 lanes -- instead of one LDS round trip
            //  per 64 cycles inside the chain)
            const uint32_t h0 = sm.cyc_group[lane], h1 = sm.cyc_group[lane + 64];
            const uint32_t cnt = uint32_t(__popcll(__ballot(lane < t && h0 == g0))) + uint32_t(__popcll(__ballot(lane + 64 < t && h1 == g0)));
            const GroupInfo &gi = sm.grp[r_tail ? 1 : 0][ridx];
            if (cnt < gi.n) {
                pcol = gi.col[cnt]; pd = gi.d[cnt];
            } else {
                pcol = gi.colx; pd = gi.dx;                  // every kept column of the group is taken: all metrics equal
            }
            // a visit consumes a kept column only while one is left: the fallback does not
            if (lane == 0) sm.cyc_group[t] = cnt < gi.n ? g0 : 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            const bool p_tail = err < 0.f;
            const uint32_t pos = p_tail ? tpos : hpos;
            const uint32_t first = p_tail ? PP : NP;
            if (pos >= first && pos < first + Z) {
                pcol = k0col; pd = k0d;
            } else {
                const bool from_low = pos < first;
                const bool use_neg = p_tail ? !from_low : from_low;
                const int li = use_neg ? (from_low ? 2 : 3) : (from_low ? 4 : 5);
                const uint32_t ptr = use_neg ? (from_low ? a_nlo : a_nhi) : (from_low ? a_plo : a_phi);
                const uint32_t taken = use_neg ? a_nlo + a_nhi : a_plo + a_phi;       // from either end of that pool
                const uint32_t pool = use_neg ? NP : PP;
                if (taken < pool && ptr < sm.nlist[li]) {
                    const ListEntry e = sm.list[li][ptr];
                    pcol = e.col; pd = e.d;
                    a_nlo += (use_neg && from_low) ? 1u : 0u;
                    a_nhi += (use_neg && !from_low) ? 1u : 0u;
                    a_plo += (!use_neg && from_low) ? 1u : 0u;
                    a_phi += (!use_neg && !from_low) ? 1u : 0u;
                }
            }
            if (p_tail) ++tpos; else ++hpos;
        }
        const float after = ieee_add(ieee_add(err, pd), -rd);
        const float sa = after > 0.f ? 1.f : (after < 0.f ? -1.f : 0.f);
        bool un = u && (fabsf(err) > thr);
        if (NM || !without_same_sign) un = un && (sign0 == sa);
        u = un;
        if (!u && stop == 0x7FFFFFFF) stop = t + 1;
        if (tid == 0) events[row * max_cycle + t] = (pcol & 0x3FFFu) | ((rcol & 0x3FFFu) << 14) | (u ? 1u << 28 : 0u);
        if (u) {
            err = ieee_add(err, pd);
            err = ieee_add(err, -rd);
        }
    }
    if (tid == 0) t_row[row] = stop;
}

template <typename T, bool NM>
static int lists_dispatch(const void *W, int64_t out_f, int64_t in_f, int64_t ldw, const uint8_t *keep0, const float *sq,
                          const float *sum_row, const float *var_row, int use_wanda_init, int prune_m, int max_cycle,
                          float thr, float pow_var, int without_same_sign, uint32_t *events, int32_t *t_row, hipStream_t st) {
    using raw = typename T::raw;
    const int64_t nchunks = in_f / 8;
    // smallest workgroup that holds the row with <= 4 chunks per lane (fewest barriers per radix pass)
    int nw = 1;
    while (nw < 8 && nchunks > int64_t(64) * nw * 4) nw *= 2;
    if (nw == 1 && nchunks > 128) nw = 2;   // 129..256 chunks: two waves with 2 chunks per lane (one wave with 4: 174 VGPRs, 2 waves per SIMD; 6144 x 1408: 339 -> 277 us)
    if (NM && nw == 1) nw = 2;      // n:m: 23 KB of LDS per row -> six one-wave workgroups per CU would leave the SIMDs at 1.5 waves
    if (const char *e = getenv("VLMC_DSNOT_NW")) {
        const int f = atoi(e);
        if ((f == 1 || f == 2 || f == 4 || f == 8) && nchunks <= int64_t(64) * f * 4 && !(NM && f == 1)) nw = f;
    }
    if (nchunks > int64_t(64) * nw * 4) return VLMC_EINVAL;
    const int ch = nchunks <= int64_t(64) * nw * 2 ? 2 : 4;
    const char *radix_only = getenv("VLMC_DSNOT_RADIX_ONLY");      // the exact route alone (tests compare the two)
    const int fast = (radix_only && atoi(radix_only)) ? 0 : 1;
#define VLMC_DL(CH, NW)                                                                                                    \
    hipLaunchKernelGGL((dsnot_lists_kernel<T, CH, NW, NM>), dim3(unsigned(out_f)), dim3(64 * NW), 0, st,                    \
                       static_cast<const raw *>(W), out_f, in_f, ldw, keep0, sq, sum_row, var_row, use_wanda_init, prune_m, \
                       max_cycle, thr, pow_var, without_same_sign, events, t_row, fast)
#define VLMC_DL_NW(NW) do { if (ch == 2) VLMC_DL(2, NW); else VLMC_DL(4, NW); } while (0)
    switch (nw) {
        case 1:
            // (one wave per row exists for rows of <= 128 chunks only -- with 4 chunks per lane it compiled to 2 waves per
            // SIMD and two waves per row are faster: above -- and never for n:m)
            if constexpr (!NM) {
                if (ch == 2) VLMC_DL(2, 1);
                else VLMC_DL(4, 2);
            } else {
                return VLMC_EINVAL;
            }
            break;
        case 2: VLMC_DL_NW(2); break;
        case 4: VLMC_DL_NW(4); break;
        default: VLMC_DL_NW(8); break;
    }
#undef VLMC_DL_NW
#undef VLMC_DL
    return VLMC_OK;
}

template <bool NM>
static int lists_dispatch_dtype(const void *W, int dtype, int64_t out_f, int64_t in_f, int64_t ldw, const uint8_t *keep0,
                                const float *sq, const float *sum_row, const float *var_row, int use_wanda_init, int prune_m,
                                int max_cycle, float thr, float pow_var, int without_same_sign, uint32_t *events, int32_t *t_row,
                                hipStream_t st) {
    switch (dtype) {
        case VLMC_F32: return lists_dispatch<f32_t, NM>(W, out_f, in_f, ldw, keep0, sq, sum_row, var_row, use_wanda_init, prune_m,
                                                        max_cycle, thr, pow_var, without_same_sign, events, t_row, st);
        case VLMC_F16: return lists_dispatch<f16_t, NM>(W, out_f, in_f, ldw, keep0, sq, sum_row, var_row, use_wanda_init, prune_m,
                                                        max_cycle, thr, pow_var, without_same_sign, events, t_row, st);
        case VLMC_BF16: return lists_dispatch<bf16_t, NM>(W, out_f, in_f, ldw, keep0, sq, sum_row, var_row, use_wanda_init, prune_m,
                                                          max_cycle, thr, pow_var, without_same_sign, events, t_row, st);
    }
    return VLMC_EINVAL;
}

}  // namespace vlmc
