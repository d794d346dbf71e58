// K11-K13: DSnoT statistics and mask refinement
// (replaces /root/reference/lavis/compression/pruners/dsnot_pruner.py:79-101 and the cycle loops
// of :407-552 (n:m) and :553-751 (unstructured) -- ~20 small kernels per cycle x 100 cycles there).
//
// Statistics (K11): per hook call and channel, in one pass over the tokens: the squared norm (as
// K1), the plain sum (sequential fp32 adds) and the population variance (Welford in fp64, rounded
// to fp32 once -- torch's CPU var accumulates in double).  `dsnot_stats_update` then applies the
// reference's three running means in call order.
//
// Refinement (K12/K13): one workgroup per row keeps, per column, the signed metric D = W*sum_row,
// the regrowing key G = (pruned ? D : 0) / var^p and the Wanda key in registers.  The reference
// walks two sorted lists with head/tail pointers; since a pointer only ever moves one step per
// cycle, "the next element from the head / tail" is a running arg-min / arg-max over the
// not-yet-visited columns (DPP reduction, ties broken by column exactly like the stable sorts),
// so no list is ever sorted.  All rows run the same number of cycles in the reference (its loop
// stops when NO row updates any more), so the kernel records every row's (p, r, update) events
// for max_cycle cycles plus the cycle at which the row stopped updating; `dsnot_apply` replays
// the first C = min(max_cycle, max_row stop) events into the mask.  Literal semantics kept: in
// the unstructured branch every cycle ends with mask[p] = keep, mask[r] = pruned whatever the
// update flag says.
#include <cstdlib>

#include "common.hpp"
#include "topk_order.hpp"

namespace vlmc {

// ------------------------------------------------------------------------------------------
// statistics
// ------------------------------------------------------------------------------------------
// One lane = one channel of one call, tokens in order: the squared norm is the same fp32 fma chain as
// act_sqnorm_kernel (bit-exact scaler_row), the sum the same sequential fp32 adds; the variance comes from fp64
// sums of x and x^2 (full-rate on CDNA, no division per token), which agrees with torch.var to ~1e-12 relative --
// the reference's own reduction order is not reproducible to the bit anyway (tests: rtol 2e-6).
// HBM-bound streaming: 16 tokens in flight per lane through non-temporal loads.
template <typename T>
__global__ __launch_bounds__(64) void act_moments_kernel(const typename T::raw *__restrict__ x, int64_t tokens, int64_t in_f,
                                                         int64_t row_stride, int64_t call_stride, float *__restrict__ normsq,
                                                         float *__restrict__ sums, float *__restrict__ vars) {
    const int64_t ch = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (ch >= in_f) return;
    const int64_t call = blockIdx.y;
    const typename T::raw *p = x + call * call_stride + ch;
    float sq = 0.f, s = 0.f;
    double s1 = 0.0, s2 = 0.0;
    constexpr int U = 16;
    int64_t t = 0;
    for (; t + U <= tokens; t += U) {
        typename T::raw r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = __builtin_nontemporal_load(p + (t + u) * row_stride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float v = to_f32<T>(r[u]);
            sq = __builtin_fmaf(v, v, sq);
            s = ieee_add(s, v);
            const double d = double(v);
            s1 += d;
            s2 = __builtin_fma(d, d, s2);
        }
    }
    for (; t < tokens; ++t) {
        const float v = to_f32<T>(p[t * row_stride]);
        sq = __builtin_fmaf(v, v, sq);
        s = ieee_add(s, v);
        const double d = double(v);
        s1 += d;
        s2 = __builtin_fma(d, d, s2);
    }
    const int64_t o = call * in_f + ch;
    if (normsq) { const float r = ieee_sqrt(sq); normsq[o] = ieee_mul(r, r); }
    if (sums) sums[o] = s;
    if (vars) {                                                   // torch.var(unbiased=False)
        const double n = double(tokens), mean = s1 / n;
        double var = s2 / n - mean * mean;
        vars[o] = float(var > 0.0 ? var : 0.0);
    }
}

// scaler_row / sum_metric_row: *= float(n/(n+b)); += v / float(n+b)      (dsnot_pruner.py:96-101)
// var: first call -> var_c; else (var*ntok + var_c*num) / (ntok + num)    (:92)
__global__ void dsnot_stats_update_kernel(float *__restrict__ scaler, float *__restrict__ sum_row, float *__restrict__ var_row,
                                          int64_t in_f, int64_t n0, int64_t ntok0, const float *__restrict__ normsq,
                                          const float *__restrict__ sums, const float *__restrict__ vars,
                                          const int64_t *__restrict__ tokens, int64_t n_calls, int64_t batch,
                                          float *__restrict__ sqrt_out) {
    const int64_t ch = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (ch >= in_f) return;
    const bool fresh = n0 == 0 && n_calls > 0;
    float a = fresh ? 0.f : scaler[ch], b = fresh ? 0.f : sum_row[ch], v = var_row[ch];
    int64_t n = n0, ntok = ntok0;
    for (int64_t c = 0; c < n_calls; ++c) {
        const float f = float(double(n) / double(n + batch));
        const int64_t num = tokens[c];
        const float vc = vars[c * in_f + ch];
        if (ntok == 0) {
            v = vc;
        } else {
            const float t1 = ieee_mul(v, float(ntok)), t2 = ieee_mul(vc, float(num));
            v = ieee_div(ieee_add(t1, t2), float(ntok + num));
        }
        ntok += num;
        n += batch;
        a = ieee_add(ieee_mul(a, f), ieee_div(normsq[c * in_f + ch], float(n)));
        b = ieee_add(ieee_mul(b, f), ieee_div(sums[c * in_f + ch], float(n)));
    }
    scaler[ch] = a;
    sum_row[ch] = b;
    var_row[ch] = v;
    if (sqrt_out) sqrt_out[ch] = ieee_sqrt(a);
}

// ------------------------------------------------------------------------------------------
// refinement
// ------------------------------------------------------------------------------------------
// order-preserving key of a signed float for ascending sorts: -0 == +0, NaN last
__device__ __forceinline__ uint32_t signed_key(float x) {
    if (x != x) return 0xFFFFFFFFu;
    x = x + 0.f;                                             // -0 -> +0
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct Best {          // candidate of an arg-min / arg-max: key, column, payload (D of that column)
    uint32_t key, col;
    float d;
};
// lexicographic (key, col): `take_min` -> smaller wins; else larger wins.  col == 0xFFFFFFFF marks "none".
__device__ __forceinline__ bool better(const Best &a, const Best &b, bool take_min) {
    if (b.col == 0xFFFFFFFFu) return true;
    if (a.col == 0xFFFFFFFFu) return false;
    if (a.key != b.key) return take_min ? a.key < b.key : a.key > b.key;
    return take_min ? a.col < b.col : a.col > b.col;
}
template <int CTRL> __device__ __forceinline__ Best dpp_pick(Best v, bool take_min) {
    Best o;
    o.key = uint32_t(__builtin_amdgcn_update_dpp(int(v.key), int(v.key), CTRL, 0xF, 0xF, false));
    o.col = uint32_t(__builtin_amdgcn_update_dpp(int(v.col), int(v.col), CTRL, 0xF, 0xF, false));
    o.d = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v.d), __float_as_int(v.d), CTRL, 0xF, 0xF, false));
    return better(v, o, take_min) ? v : o;
}
__device__ __forceinline__ Best wave_best(Best v, bool take_min) {
    v = dpp_pick<0xB1>(v, take_min);       // quad_perm [1,0,3,2]
    v = dpp_pick<0x4E>(v, take_min);       // quad_perm [2,3,0,1]
    v = dpp_pick<0x141>(v, take_min);      // row_half_mirror
    v = dpp_pick<0x140>(v, take_min);      // row_mirror
    // every lane of a row holds the row's best; pick the winning row on (key, col) alone and fetch its payload
    // afterwards (selecting the float payload together with the integer fields was miscompiled by hipcc 7.2
    // in the multi-wave instantiations: the payload of the first candidate survived)
    uint32_t bk = uint32_t(__builtin_amdgcn_readlane(int(v.key), 0)), bc = uint32_t(__builtin_amdgcn_readlane(int(v.col), 0));
    int win = 0;
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const uint32_t k = uint32_t(__builtin_amdgcn_readlane(int(v.key), r * 16)),
                       c = uint32_t(__builtin_amdgcn_readlane(int(v.col), r * 16));
        const bool bt = better(Best{k, c, 0.f}, Best{bk, bc, 0.f}, take_min);
        bk = bt ? k : bk;
        bc = bt ? c : bc;
        win = bt ? r * 16 : win;
    }
    return Best{bk, bc, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.d), win))};
}

template <int NW> struct DsSmem {
    uint32_t key[2][NW], col[2][NW];
    float d[2][NW], fsum[NW];
};

template <int NW>
__device__ __forceinline__ Best block_best(Best v, bool take_min, DsSmem<NW> &sm, int &phase) {
    v = wave_best(v, take_min);
    if constexpr (NW > 1) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sm.key[phase][wave] = v.key; sm.col[phase][wave] = v.col; sm.d[phase][wave] = v.d; }
        __syncthreads();
        uint32_t bk = sm.key[phase][0], bc = sm.col[phase][0];
        int win = 0;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const uint32_t k = sm.key[phase][w], c = sm.col[phase][w];
            const bool bt = better(Best{k, c, 0.f}, Best{bk, bc, 0.f}, take_min);
            bk = bt ? k : bk;
            bc = bt ? c : bc;
            win = bt ? w : win;
        }
        const Best best{bk, bc, sm.d[phase][win]};   // payload fetched by index (see wave_best)
        phase ^= 1;
        v = best;
    }
    return v;
}

// One workgroup per row.  events[row, t] = p | r << 14 | u << 28 for cycle t (0-based).
template <typename T, int CH, int NW, bool NM>
__global__ __launch_bounds__(64 * NW) void dsnot_simulate_kernel(
    const typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f, int64_t ldw, const uint8_t *__restrict__ keep0,
    const float *__restrict__ sqrt_scaler, const float *__restrict__ sum_row, const float *__restrict__ var_row, int use_wanda_init,
    int prune_m, int max_cycle, float thr, float pow_var, int without_same_sign, uint32_t *__restrict__ events,
    int32_t *__restrict__ t_row) {
    constexpr int NT = 64 * NW, E = CH * 8;
    __shared__ DsSmem<NW> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int64_t nchunks = in_f / 8;
    float D[E];
    uint32_t gk[E], wk[E];
    uint32_t live = 0, pruned0 = 0;          // bit i: column exists / initially pruned
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        const int64_t c = int64_t(s) * NT + tid;
        if (c < nchunks) {
            const int64_t col0 = c * 8;
            Chunk8<T> raw = load_chunk8<T>(W + row * ldw + col0);
            const uint2 mm = *reinterpret_cast<const uint2 *>(keep0 + row * in_f + col0);
            uint8_t m[8];
            __builtin_memcpy(m, &mm, 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = s * 8 + j;
                const float w = to_f32<T>(raw.v[j]);
                const float d = ieee_mul(w, sum_row[col0 + j]);          // signed weight * mean activation (:384)
                D[i] = d;
                const bool pr = m[j] == 0;
                live |= 1u << i;
                if (pr) pruned0 |= 1u << i;
                // metric used to order the kept columns / pick the group minimum
                const float init = use_wanda_init ? ieee_mul(fabsf(w), sqrt_scaler[col0 + j]) : fabsf(w);
                const float wanda = NM ? init : ieee_mul(fabsf(w), sqrt_scaler[col0 + j]);
                wk[i] = score_key(wanda);
                float g = pr ? d : 0.f;
                if (pr) part = ieee_add(part, d);
                if (pow_var != 0.f) {
                    const float v = var_row[col0 + j];
                    g = ieee_div(g, pow_var == 1.f ? v : powf(v, pow_var));
                }
                gk[i] = signed_key(g);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { D[s * 8 + j] = 0.f; gk[s * 8 + j] = 0; wk[s * 8 + j] = 0; }
        }
    }
    // reconstruction error = sum of D over the pruned columns (fixed tree: lane-sequential, DPP, waves in order)
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
        }
    }
    const float sign0 = err > 0.f ? 1.f : (err < 0.f ? -1.f : 0.f);

    // visited sets: regrow list from the head / tail; prune list: negatives from the low / high end,
    // positives from the low / high end (see prune_list layout below)
    uint32_t seen_r = 0, neg_lo = 0, neg_hi = 0, pos_lo = 0, pos_hi = 0;
    uint32_t kept_now = live & ~pruned0;                     // n:m: columns currently eligible as the group minimum
    // unstructured prune list = [kept with D<0 by ascending wanda | Z x K0 | kept with D>0 by DESCENDING wanda]
    uint32_t negm = 0, posm = 0;
    uint32_t NP = 0, PP = 0, Z = 0, k0col = 0;
    float k0d = 0.f;
    int phase = 0;
    if constexpr (!NM) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            if ((kept_now >> i) & 1u) {
                if (D[i] < 0.f) negm |= 1u << i;
                else if (D[i] > 0.f) posm |= 1u << i;
            }
        }
        const uint32_t a = wave_sum_u32(uint32_t(__popc(negm))), b = wave_sum_u32(uint32_t(__popc(posm))),
                       c = wave_sum_u32(uint32_t(__popc(kept_now)));
        NP = a; PP = b; Z = c;
        if constexpr (NW > 1) {
            __syncthreads();
            if (lane == 0) { sm.key[0][wave] = a; sm.col[0][wave] = b; sm.key[1][wave] = c; }
            __syncthreads();
            NP = PP = Z = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { NP += sm.key[0][w]; PP += sm.col[0][w]; Z += sm.key[1][w]; }
            __syncthreads();
        }
        Z = Z - NP - PP;                                     // kept columns with D == 0
        Best k0{0, 0xFFFFFFFFu, 0.f};                        // kept column with the smallest wanda metric
#pragma unroll
        for (int i = 0; i < E; ++i)
            if ((kept_now >> i) & 1u) {
                const Best c2{wk[i], uint32_t((i / 8) * NT * 8 + tid * 8 + (i % 8)), D[i]};
                if (better(c2, k0, true)) k0 = c2;
            }
        k0 = block_best<NW>(k0, true, sm, phase);
        k0col = k0.col; k0d = k0.d;
    }

    // Every lane caches its own best candidate of each sorted list the cycles walk (smallest / largest unvisited G;
    // smallest / largest unconsumed wanda metric among its negative-D and positive-D kept columns).  A cycle is then
    // two workgroup-wide arg-min reductions over the cached candidates; only the ONE lane whose candidate was taken
    // rescans its <= 32 columns.  (Rescanning every column in every lane cost 40-70 x more per cycle.)
    auto scan = [&](const uint32_t (&keys)[E], uint32_t pool, bool take_min) -> Best {
        Best bst{0, 0xFFFFFFFFu, 0.f};
#pragma unroll
        for (int i = 0; i < E; ++i)
            if ((pool >> i) & 1u) {
                const Best c2{keys[i], uint32_t((i / 8) * NT * 8 + tid * 8 + (i % 8)), D[i]};
                if (better(c2, bst, take_min)) bst = c2;
            }
        return bst;
    };
    Best c_rlo = scan(gk, live, true), c_rhi = scan(gk, live, false);
    Best c_nlo{0, 0xFFFFFFFFu, 0.f}, c_nhi = c_nlo, c_plo = c_nlo, c_phi = c_nlo;
    if constexpr (!NM) {
        c_nlo = scan(wk, negm, true); c_nhi = scan(wk, negm, false);
        c_plo = scan(wk, posm, true); c_phi = scan(wk, posm, false);
    }

    bool u = true;
    int stop = 0x7FFFFFFF;
    uint32_t hpos = 0, tpos = 0;                             // prune-list positions consumed from head / tail
    for (int t = 0; t < max_cycle; ++t) {
        // ---- regrow candidate: next from the tail (err > 0) or the head of the ascending G order ----
        const bool r_tail = err > 0.f;
        Best rb = block_best<NW>(r_tail ? c_rhi : c_rlo, !r_tail, sm, phase);
        const uint32_t rcol = rb.col;
        {   // the owner marks it visited and refreshes its cached candidates
            const uint32_t c = rcol / 8, own = c % NT, slot = c / NT;
            if (uint32_t(tid) == own) {
                seen_r |= 1u << (slot * 8 + rcol % 8);
                if (c_rlo.col == rcol) c_rlo = scan(gk, live & ~seen_r, true);     // only the list(s) that lost their head
                if (c_rhi.col == rcol) c_rhi = scan(gk, live & ~seen_r, false);
            }
        }
        // ---- prune candidate ---------------------------------------------------------------------
        Best pb{0, 0xFFFFFFFFu, 0.f};
        if constexpr (NM) {
            // `torch.topk(pruning_block, 1, largest=False)` over r's m-group (:517-519): kept columns carry their metric,
            // pruned and already taken ones +inf; equal minima (an exhausted group is all +inf) are decided as the
            // reference's CPU run decides them (topk_order.hpp)
            const uint32_t g0 = rcol - rcol % uint32_t(prune_m);
            if (uint32_t(tid) == (rcol / 8) % NT) {          // the m-group lies inside one lane's 8-column chunk
                uint32_t gkey[8];
                float gd[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) { gkey[a] = 0x7F800000u; gd[a] = 0.f; }
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const uint32_t col = uint32_t((i / 8) * NT * 8 + tid * 8 + (i % 8));
                    if (col >= g0 && col < g0 + uint32_t(prune_m)) {
#pragma unroll
                        for (int a = 0; a < 8; ++a)
                            if (uint32_t(a) == col - g0) {
                                gkey[a] = ((kept_now >> i) & 1u) ? wk[i] : 0x7F800000u;
                                gd[a] = D[i];
                            }
                    }
                }
                const int pick = torch_cpu_argmin(gkey, prune_m);
                float pd = 0.f;
#pragma unroll
                for (int a = 0; a < 8; ++a) pd = a == pick ? gd[a] : pd;
                pb = Best{0u, g0 + uint32_t(pick), pd};
            }
            pb = block_best<NW>(pb, true, sm, phase);
        } else {
            const bool p_tail = err < 0.f;
            const uint32_t pos = p_tail ? tpos : hpos;       // steps already taken from that end
            // head: [0,NP) negatives ascending | [NP,NP+Z) K0 | then positives descending
            // tail: [0,PP) positives ascending | [PP,PP+Z) K0 | then negatives descending
            const uint32_t first = p_tail ? PP : NP;
            if (pos >= first && pos < first + Z) {
                pb.col = k0col; pb.d = k0d;
            } else {
                const bool from_low = pos < first;           // still inside the own-sign run
                // head&low: negatives ascending (neg_lo); head&high: positives descending (pos_hi)
                // tail&low: positives ascending (pos_lo); tail&high: negatives descending (neg_hi)
                const bool use_neg = p_tail ? !from_low : from_low;
                pb = block_best<NW>(use_neg ? (from_low ? c_nlo : c_nhi) : (from_low ? c_plo : c_phi), from_low, sm, phase);
                const uint32_t c = pb.col / 8, own = c % NT, slot = c / NT;
                if (pb.col != 0xFFFFFFFFu && uint32_t(tid) == own) {
                    const uint32_t bit = 1u << (slot * 8 + pb.col % 8);
                    if (use_neg) {
                        if (from_low) neg_lo |= bit; else neg_hi |= bit;
                        if (c_nlo.col == pb.col) c_nlo = scan(wk, negm & ~(neg_lo | neg_hi), true);
                        if (c_nhi.col == pb.col) c_nhi = scan(wk, negm & ~(neg_lo | neg_hi), false);
                    } else {
                        if (from_low) pos_lo |= bit; else pos_hi |= bit;
                        if (c_plo.col == pb.col) c_plo = scan(wk, posm & ~(pos_lo | pos_hi), true);
                        if (c_phi.col == pb.col) c_phi = scan(wk, posm & ~(pos_lo | pos_hi), false);
                    }
                }
            }
            if (p_tail) ++tpos; else ++hpos;
        }
        // ---- update rule ----------------------------------------------------------------------------
        const float after = ieee_add(ieee_add(err, pb.d), -rb.d);          // err + D[p] - D[r]
        const float sa = after > 0.f ? 1.f : (after < 0.f ? -1.f : 0.f);
        bool un = u && (fabsf(err) > thr);
        if (NM || !without_same_sign) un = un && (sign0 == sa);
        u = un;
        if (!u && stop == 0x7FFFFFFF) stop = t + 1;
        if (tid == 0) events[row * max_cycle + t] = (pb.col & 0x3FFFu) | ((rcol & 0x3FFFu) << 14) | (u ? 1u << 28 : 0u);
        if constexpr (NM) {
            // the reference marks p's metric as taken in every cycle (:531), whatever `u` says
            const uint32_t c = pb.col / 8, own = c % NT, slot = c / NT;
            if (uint32_t(tid) == own) kept_now &= ~(1u << (slot * 8 + pb.col % 8));
        }
        if (u) {
            err = ieee_add(err, pb.d);
            err = ieee_add(err, -rb.d);
        }
    }
    if (tid == 0) t_row[row] = stop;
}

// Replays the first C events of every row into the keep mask and (optionally) zeroes the pruned weights.
template <typename T>
__global__ __launch_bounds__(256) void dsnot_apply_kernel(typename T::raw *__restrict__ W, int64_t out_f, int64_t in_f, int64_t ldw,
                                                          uint8_t *__restrict__ keep, const uint32_t *__restrict__ events,
                                                          const int32_t *__restrict__ ncycles, int max_cycle, int nm_mode,
                                                          int apply_zero) {
    extern __shared__ __attribute__((aligned(16))) uint8_t row_keep[];     // [in_f rounded up to 16] mask bytes, then max_cycle events
    const int64_t row = blockIdx.x;
    uint32_t *ev = reinterpret_cast<uint32_t *>(row_keep + ((in_f + 15) & ~int64_t(15)));
    int C = *ncycles;
    if (C > max_cycle) C = max_cycle;
    // 16 mask bytes (and 16 weights) per lane when the row allows it: the byte-wise version moved 64 B per wave-instruction
    const bool vec = (in_f & 15) == 0 && (ldw & 7) == 0 && (reinterpret_cast<uintptr_t>(keep) & 15u) == 0 &&
                     (reinterpret_cast<uintptr_t>(W) & 15u) == 0 && sizeof(typename T::raw) == 2;
    if (vec) {
        for (int64_t c = int64_t(threadIdx.x) * 16; c < in_f; c += int64_t(blockDim.x) * 16)
            *reinterpret_cast<u32x4_t *>(row_keep + c) = *reinterpret_cast<const u32x4_t *>(keep + row * in_f + c);
    } else {
        for (int64_t c = threadIdx.x; c < in_f; c += blockDim.x) row_keep[c] = keep[row * in_f + c];
    }
    for (int t = threadIdx.x; t < C; t += blockDim.x) ev[t] = events[row * max_cycle + t];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 0; t < C; ++t) {
            const uint32_t e = ev[t];
            const uint32_t p = e & 0x3FFFu, r = (e >> 14) & 0x3FFFu, u = (e >> 28) & 1u;
            if (nm_mode) {
                row_keep[p] = u ? 0 : 1;      // weight_mask[p] = update_mask   (pruned <=> keep = 0)  (:533)
                row_keep[r] = u ? 1 : 0;      // weight_mask[r] = ~update_mask                          (:534)
            } else {
                row_keep[p] = 1;              // net effect of :727-740: p un-pruned ...
                row_keep[r] = 0;              // ... r pruned, in every cycle
            }
        }
    }
    __syncthreads();
    if (vec) {
        for (int64_t c = int64_t(threadIdx.x) * 16; c < in_f; c += int64_t(blockDim.x) * 16) {
            const u32x4_t kv = *reinterpret_cast<const u32x4_t *>(row_keep + c);
            *reinterpret_cast<u32x4_t *>(keep + row * in_f + c) = kv;
            if (apply_zero) {
                uint8_t kb[16];
                __builtin_memcpy(kb, &kv, 16);
                bool all = true;
#pragma unroll
                for (int j = 0; j < 16; ++j) all = all && kb[j] != 0;
                if (!all) {
                    typename T::raw *wp = W + row * ldw + c;
                    u32x4_t w0 = *reinterpret_cast<const u32x4_t *>(wp), w1 = *reinterpret_cast<const u32x4_t *>(wp + 8);
                    uint16_t e[16];
                    __builtin_memcpy(e, &w0, 16);
                    __builtin_memcpy(e + 8, &w1, 16);
#pragma unroll
                    for (int j = 0; j < 16; ++j) e[j] = kb[j] ? e[j] : uint16_t(0);
                    __builtin_memcpy(&w0, e, 16);
                    __builtin_memcpy(&w1, e + 8, 16);
                    *reinterpret_cast<u32x4_t *>(wp) = w0;
                    *reinterpret_cast<u32x4_t *>(wp + 8) = w1;
                }
            }
        }
        return;
    }
    for (int64_t c = threadIdx.x; c < in_f; c += blockDim.x) {
        const uint8_t k = row_keep[c];
        keep[row * in_f + c] = k;
        if (apply_zero && !k) W[row * ldw + c] = typename T::raw(0);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <typename T, bool NM>
static int simulate_dispatch(const void *W, int64_t out_f, int64_t in_f, int64_t ldw, const uint8_t *keep0, const float *sq,
                             const float *sum_row, const float *var_row, int use_wanda_init, int prune_m, int max_cycle,
                             float thr, float pow_var, int without_same_sign, uint32_t *events, int32_t *t_row, hipStream_t st) {
    using raw = typename T::raw;
    const int64_t nchunks = in_f / 8;
    // A cycle costs the rescan of ONE lane's columns plus two workgroup-wide reductions (with a barrier each when the
    // row spans several waves).  Measured (100 cycles, fp16): rows of 1408 columns are fastest in one wave (1.2 ms per
    // 6144x1408 linear vs 1.5 / 2.0 / 4.9 ms with 2 / 4 / 8 waves), rows of 4096 columns with four (2.2 ms vs 3.2 ms
    // with two and 4.0 ms with eight).
    int nw = nchunks <= 256 ? 1 : (nchunks <= 768 ? 4 : 8);
    if (const char *e = getenv("VLMC_DSNOT_NW")) {                    // tuning override
        const int f = atoi(e);
        if ((f == 1 || f == 2 || f == 4 || f == 8) && nchunks <= int64_t(64) * f * 4) nw = f;
    }
    if (nchunks > int64_t(64) * nw * 4) {
        set_error("vlmc_dsnot_refine: in_features %lld too large (max 16384)", (long long)in_f);
        return VLMC_EINVAL;
    }
    const int ch = int((nchunks + 64 * nw - 1) / (64 * nw));
#define VLMC_SIM(CH, NW)                                                                                                  \
    hipLaunchKernelGGL((dsnot_simulate_kernel<T, CH, NW, NM>), dim3(unsigned(out_f)), dim3(64 * NW), 0, st,                \
                       static_cast<const raw *>(W), out_f, in_f, ldw, keep0, sq, sum_row, var_row, use_wanda_init, prune_m, \
                       max_cycle, thr, pow_var, without_same_sign, events, t_row)
#define VLMC_SIM_NW(NW)                     \
    switch (ch) {                           \
        case 1: VLMC_SIM(1, NW); break;     \
        case 2: VLMC_SIM(2, NW); break;     \
        case 3: VLMC_SIM(3, NW); break;     \
        default: VLMC_SIM(4, NW); break;    \
    }
    switch (nw) {
        case 1: VLMC_SIM_NW(1); break;
        case 2: VLMC_SIM_NW(2); break;
        case 4: VLMC_SIM_NW(4); break;
        default: VLMC_SIM_NW(8); break;
    }
#undef VLMC_SIM_NW
#undef VLMC_SIM
    VLMC_HIP_CHECK_LAUNCH("vlmc_dsnot_refine");
    return VLMC_OK;
}

}  // namespace vlmc

namespace vlmc {
int dsnot_refine_lists(const void *W, int dtype, int64_t out_f, int64_t in_f, int64_t ldw, const uint8_t *keep0, const float *sq,
                       const float *sum_row, const float *var_row, int use_wanda_init, int prune_n, int prune_m, int max_cycle,
                       float thr, float pow_var, int without_same_sign, uint32_t *events, int32_t *t_row, hipStream_t st);
}
using namespace vlmc;

extern "C" int vlmc_act_moments(const void *x, int dtype, int64_t n_calls, int64_t tokens, int64_t in_features,
                                int64_t row_stride, int64_t call_stride, float *normsq, float *sums, float *vars, void *stream) {
    VLMC_REQUIRE(x && (normsq || sums || vars), "vlmc_act_moments: null pointer");
    VLMC_REQUIRE(n_calls >= 0 && n_calls <= 65535 && tokens > 0 && in_features > 0 && row_stride >= in_features,
                 "vlmc_act_moments: bad shape calls=%lld tokens=%lld in=%lld", (long long)n_calls, (long long)tokens,
                 (long long)in_features);
    if (n_calls == 0) return VLMC_OK;
    const dim3 grid(unsigned((in_features + 63) / 64), unsigned(n_calls));
    hipStream_t st = as_stream(stream);
#define VLMC_MOM(T) hipLaunchKernelGGL((act_moments_kernel<T>), grid, dim3(64), 0, st, static_cast<const T::raw *>(x), tokens, \
                                       in_features, row_stride, call_stride, normsq, sums, vars)
    switch (dtype) {
        case VLMC_F32: VLMC_MOM(f32_t); break;
        case VLMC_F16: VLMC_MOM(f16_t); break;
        case VLMC_BF16: VLMC_MOM(bf16_t); break;
        default: set_error("vlmc_act_moments: unknown dtype %d", dtype); return VLMC_EINVAL;
    }
#undef VLMC_MOM
    VLMC_HIP_CHECK_LAUNCH("vlmc_act_moments");
    return VLMC_OK;
}

extern "C" int vlmc_dsnot_stats_update(float *scaler_row, float *sum_row, float *var_row, int64_t in_features,
                                       int64_t nsamples_before, int64_t ntokens_before, const float *normsq, const float *sums,
                                       const float *vars, const int64_t *tokens_per_call, int64_t n_calls, int64_t batch,
                                       float *sqrt_out, void *stream) {
    VLMC_REQUIRE(scaler_row && sum_row && var_row && (n_calls == 0 || (normsq && sums && vars && tokens_per_call)),
                 "vlmc_dsnot_stats_update: null pointer");
    VLMC_REQUIRE(in_features > 0 && n_calls >= 0 && batch > 0, "vlmc_dsnot_stats_update: bad arguments");
    hipLaunchKernelGGL(dsnot_stats_update_kernel, dim3(unsigned((in_features + 255) / 256)), dim3(256), 0, as_stream(stream),
                       scaler_row, sum_row, var_row, in_features, nsamples_before, ntokens_before, normsq, sums, vars,
                       tokens_per_call, n_calls, batch, sqrt_out);
    VLMC_HIP_CHECK_LAUNCH("vlmc_dsnot_stats_update");
    return VLMC_OK;
}

extern "C" int vlmc_dsnot_refine(const void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw,
                                 const uint8_t *keep_mask0, const float *sqrt_scaler, const float *sum_row, const float *var_row,
                                 int use_wanda_init, int prune_n, int prune_m, int max_cycle, float update_threshold,
                                 float pow_of_var, int without_same_sign, uint32_t *events, int32_t *stop_cycle, void *stream) {
    VLMC_REQUIRE(W && keep_mask0 && sqrt_scaler && sum_row && var_row && events && stop_cycle, "vlmc_dsnot_refine: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && in_features % 8 == 0 && ldw % 8 == 0 && aligned16(W) &&
                     (reinterpret_cast<uintptr_t>(keep_mask0) % 8) == 0,
                 "vlmc_dsnot_refine: needs in_features %% 8 == 0 and 16-byte aligned rows (in=%lld)", (long long)in_features);
    VLMC_REQUIRE(max_cycle > 0 && max_cycle < in_features, "vlmc_dsnot_refine: max_cycle %d must be in (0, in_features)", max_cycle);
    if (prune_n != 0)
        VLMC_REQUIRE(prune_m == 2 || prune_m == 4 || prune_m == 8, "vlmc_dsnot_refine: n:m needs m in {2,4,8} (got %d)", prune_m);
    hipStream_t st = as_stream(stream);
    // fast path: sorted-list heads + O(1) cycles (dsnot_lists.hip); VLMC_DSNOT_LISTS=0 keeps the per-cycle reductions
    const char *use_lists = getenv("VLMC_DSNOT_LISTS");
    if (!(use_lists && atoi(use_lists) == 0)) {
        const int rc = dsnot_refine_lists(W, dtype, out_features, in_features, ldw, keep_mask0, sqrt_scaler, sum_row, var_row,
                                          use_wanda_init, prune_n, prune_m, max_cycle, update_threshold, pow_of_var,
                                          without_same_sign, events, stop_cycle, st);
        if (rc == VLMC_OK) {
            VLMC_HIP_CHECK_LAUNCH("vlmc_dsnot_refine");
            return VLMC_OK;
        }
    }
#define VLMC_DS(T)                                                                                                              \
    (prune_n != 0 ? simulate_dispatch<T, true>(W, out_features, in_features, ldw, keep_mask0, sqrt_scaler, sum_row, var_row,     \
                                               use_wanda_init, prune_m, max_cycle, update_threshold, pow_of_var,                 \
                                               without_same_sign, events, stop_cycle, st)                                        \
                  : simulate_dispatch<T, false>(W, out_features, in_features, ldw, keep_mask0, sqrt_scaler, sum_row, var_row,    \
                                                use_wanda_init, prune_m, max_cycle, update_threshold, pow_of_var,                \
                                                without_same_sign, events, stop_cycle, st))
    switch (dtype) {
        case VLMC_F32: return VLMC_DS(f32_t);
        case VLMC_F16: return VLMC_DS(f16_t);
        case VLMC_BF16: return VLMC_DS(bf16_t);
    }
#undef VLMC_DS
    set_error("vlmc_dsnot_refine: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}

extern "C" int vlmc_dsnot_apply(void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw, uint8_t *keep_mask,
                                const uint32_t *events, const int32_t *ncycles, int max_cycle, int nm_mode, int apply_zero,
                                void *stream) {
    VLMC_REQUIRE(W && keep_mask && events && ncycles, "vlmc_dsnot_apply: null pointer");
    VLMC_REQUIRE(out_features > 0 && in_features > 0 && in_features <= 16384, "vlmc_dsnot_apply: bad shape");
    hipStream_t st = as_stream(stream);
    const size_t lds = ((size_t(in_features) + 15) & ~size_t(15)) + size_t(max_cycle > 0 ? max_cycle : 1) * 4;
#define VLMC_AP(T) hipLaunchKernelGGL((dsnot_apply_kernel<T>), dim3(unsigned(out_features)), dim3(256), lds, st,              \
                                      static_cast<T::raw *>(W), out_features, in_features, ldw, keep_mask, events, ncycles,     \
                                      max_cycle, nm_mode, apply_zero)
    switch (dtype) {
        case VLMC_F32: VLMC_AP(f32_t); break;
        case VLMC_F16: VLMC_AP(f16_t); break;
        case VLMC_BF16: VLMC_AP(bf16_t); break;
        default: set_error("vlmc_dsnot_apply: unknown dtype %d", dtype); return VLMC_EINVAL;
    }
#undef VLMC_AP
    VLMC_HIP_CHECK_LAUNCH("vlmc_dsnot_apply");
    return VLMC_OK;
}

// ---- vlmc_reorder_indices: dsnot_pruner.py:1881-1925 `return_reorder_indice` as a callable of its own ---------------------------
// Per row of `x`: the column indices of the NEGATIVE entries, ascending, at the head; the indices of the POSITIVE entries,
// descending, at the tail; every position in between -- one per entry that is neither (zeros, NaN) -- holds 0.  (The reference
// forms it from two fp64 index matrices with +inf sentinels, two full sorts, a flip and a sum; the list kernel above applies the
// same rule to its kept list in LDS.)  One workgroup per row, one pass over the row: per 256-column chunk a wave ballot ranks the
// negatives and the positives, running counts carry over the chunks; the middle is zero-filled at the end.
namespace vlmc {
template <typename T>
__global__ __launch_bounds__(256) void reorder_indices_kernel(const typename T::raw *__restrict__ x, int64_t cols, int64_t ldx,
                                                              int64_t *__restrict__ out, int64_t ldo) {
    __shared__ uint32_t wneg[4], wpos[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const typename T::raw *p = x + int64_t(blockIdx.x) * ldx;
    int64_t *q = out + int64_t(blockIdx.x) * ldo;
    int64_t nneg = 0, npos = 0;                                              // entries of either sign in the chunks before this one
    for (int64_t c0 = 0; c0 < cols; c0 += 256) {
        const int64_t c = c0 + threadIdx.x;
        const float v = c < cols ? to_f32<T>(p[c]) : 0.f;
        const bool neg = v < 0.f, pos = v > 0.f;
        const uint64_t bn = __ballot(neg), bp = __ballot(pos);
        if (lane == 0) wneg[wave] = uint32_t(__popcll(bn)), wpos[wave] = uint32_t(__popcll(bp));
        __syncthreads();
        uint32_t on = 0, op = 0, tn = 0, tp = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) on += wneg[w], op += wpos[w];
            tn += wneg[w], tp += wpos[w];
        }
        const uint64_t below = lane == 0 ? 0 : (~uint64_t(0) >> (64 - lane));
        if (neg) q[nneg + on + __popcll(bn & below)] = c;
        if (pos) q[cols - 1 - (npos + op + __popcll(bp & below))] = c;
        nneg += tn, npos += tp;
        __syncthreads();
    }
    for (int64_t c = nneg + threadIdx.x; c < cols - npos; c += 256) q[c] = 0;
}
}  // namespace vlmc

extern "C" int vlmc_reorder_indices(const void *x, int dtype, int64_t rows, int64_t cols, int64_t ldx, int64_t *out, int64_t ldo,
                                    void *stream) {
    VLMC_REQUIRE(x && out, "vlmc_reorder_indices: null pointer");
    VLMC_REQUIRE(rows >= 0 && cols > 0 && ldx >= cols && ldo >= cols && rows < (int64_t(1) << 31), "vlmc_reorder_indices: bad shape");
    if (rows == 0) return VLMC_OK;
    hipStream_t st = as_stream(stream);
#define VLMC_RO(T) hipLaunchKernelGGL((reorder_indices_kernel<T>), dim3(unsigned(rows)), dim3(256), 0, st,                       \
                                      static_cast<const T::raw *>(x), cols, ldx, out, ldo)
    switch (dtype) {
        case VLMC_F32: VLMC_RO(f32_t); break;
        case VLMC_F16: VLMC_RO(f16_t); break;
        case VLMC_BF16: VLMC_RO(bf16_t); break;
        default: set_error("vlmc_reorder_indices: unknown dtype %d", dtype); return VLMC_EINVAL;
    }
#undef VLMC_RO
    VLMC_HIP_CHECK_LAUNCH("vlmc_reorder_indices");
    return VLMC_OK;
}
