// K17: threshold selection over MANY score tensors at once (the reference's global pruners).
//
// Reference: lavis/compression/pruners/global_pruner.py
//   get_mask            :107-133  (per-layer protection of the top (1-max_sparsity) share, then ONE threshold
//                                  = the int(p*N)-th smallest score over the concatenation of all layers,
//                                  masks = score > threshold)
//   get_layerwise_mask  :135-148  (the same rule, one threshold per layer)
//   global_iterative_pruning :153-201 (scores *= previous masks; weights *= masks)
//   scores: magnitude = float(w) (signed, :255), random = a given tensor (:262), aobd = |w| * |mean |g|| (:311)
//
// The reference concatenates every score on the CPU and runs torch.topk over the result.  Here nothing is
// materialised: a 3-pass radix select (11/11/10 bits of an order-preserving key) histograms the scores as they
// are recomputed from the weights, one histogram per SCOPE (all layers / one model / one layer), and a final
// pass writes keep masks and multiplies the weights.  Everything is HBM streaming: 4 reads of the operands and
// one write of W + mask (7 reads when per-layer protection is on).
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace vlmc {
namespace {

constexpr int kThreads = 256;
constexpr int kChunk = kThreads * 8;        // elements per work-group iteration
constexpr int kBins = 2048;
constexpr int kMaxWgs = 2048;

struct DevJob {
    void *W;
    const float *S;
    const uint8_t *prev;
    uint8_t *keep;
    int64_t numel;
    int64_t chunk_begin;    // first chunk id of this job in the flat chunk space
    int64_t protect_k;
    int32_t scope;
    int16_t dtype;          // VLMC_F32 / F16 / BF16: jobs of one call may differ (fp16 vision tower + bf16 language model)
    int16_t vec_ok;         // all pointers aligned for the 8-wide loads
};
struct SelState {           // one per histogram (scope, or job in the protection phase)
    uint32_t prefix;        // key bits resolved so far
    uint32_t active;
    int64_t k_rem;          // rank still to find inside the prefix (1-based)
};

// Order-preserving unsigned key of an fp32 score (-0 == +0, NaN sorts last: torch.topk treats NaN as the
// largest).  The kernels never form the key of every element: scores are first made CANONICAL
// (s + 0.0f turns -0 into +0; NaN bits become 0x7FFFFFFF), after which
//   * "key has the prefix P above bit h" is a shift + compare on the raw bits (against prefix_match(P, h)),
//   * "key > threshold" is the float comparison s > key_to_float(threshold) (false for NaN on either side).
__device__ __forceinline__ uint32_t canonical_bits(float s) {      // s already passed through s + 0.0f
    return s != s ? 0x7FFFFFFFu : __float_as_uint(s);
}
__device__ __forceinline__ uint32_t key_of_bits(uint32_t b) {      // negatives: all bits flipped, others: sign bit set
    return b ^ (uint32_t(int32_t(b) >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t key) {
    return __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}
// raw-bits constant C with: (canonical bits >> h) == C  <=>  (key >> h) == (P >> h)
__device__ __forceinline__ uint32_t prefix_match(uint32_t P, int h) {
    return ((P & 0x80000000u) ? (P ^ 0x80000000u) : ~P) >> h;
}

struct Elems {
    float w[8];             // float(w) (0 when there is no weight tensor)
    float s[8];             // canonical score (no -0)
    int n;                  // valid elements
};

template <typename T> __device__ __forceinline__ void load_w8(const void *Wv, int64_t base, int n, bool vec, float *w) {
    const typename T::raw *W = static_cast<const typename T::raw *>(Wv);
    if (vec) {
        const Chunk8<T> c = load_chunk8<T, true>(W + base);
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = to_f32<T>(c.v[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = i < n ? to_f32<T>(W[base + i]) : 0.f;
    }
}

// scores of 8 consecutive elements of a job starting at `base`
// LOADW: the weights are needed (as a score operand, or to be rewritten by the apply pass)
template <int MODE, bool LOADW> __device__ __forceinline__ void load_scores(const DevJob &j, int64_t base, Elems &e) {
    const int64_t left = j.numel - base;
    e.n = left >= 8 ? 8 : (left > 0 ? int(left) : 0);
    const bool vec = e.n == 8 && j.vec_ok;
    float s[8], p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e.w[i] = 0.f;
    if (LOADW && j.W) {
        if (j.dtype == VLMC_BF16) load_w8<bf16_t>(j.W, base, e.n, vec, e.w);
        else if (j.dtype == VLMC_F16) load_w8<f16_t>(j.W, base, e.n, vec, e.w);
        else load_w8<f32_t>(j.W, base, e.n, vec, e.w);
    }
    if (vec) {
        if (MODE != 0) {
            const Chunk8<f32_t> c = load_chunk8<f32_t, true>(j.S + base);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = c.v[i];
        }
        if (j.prev) {
            const u32x2_t q = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t *>(j.prev + base));
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = ((i < 4 ? q.x >> (8 * i) : q.y >> (8 * (i - 4))) & 0xFF) ? 1.f : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = i < e.n;
            s[i] = (ok && MODE != 0) ? j.S[base + i] : 0.f;
            p[i] = (ok && j.prev) ? (j.prev[base + i] ? 1.f : 0.f) : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float sc = MODE == 0 ? e.w[i] : MODE == 1 ? s[i] : ieee_mul(__builtin_fabsf(e.w[i]), __builtin_fabsf(s[i]));
        if (j.prev) sc = ieee_mul(sc, p[i]);          // importance *= masks (:166-169)
        e.s[i] = ieee_add(sc, 0.f);                   // -0 -> +0, everything else unchanged
    }
}

template <typename T> __device__ __forceinline__ void store_w8(void *Wv, int64_t base, int n, bool vec, const float *w) {
    using raw = typename T::raw;
    raw *W = static_cast<raw *>(Wv);
    Chunk8<T> out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (sizeof(raw) == 4) out.v[i] = w[i];
        else if constexpr (std::is_same<T, bf16_t>::value) out.v[i] = (w[i] != w[i]) ? 0x7FC0 : uint16_t(__float_as_uint(w[i]) >> 16);
        else {
            const _Float16 hh = _Float16(w[i]);
            __builtin_memcpy(&out.v[i], &hh, 2);
        }
    }
    if (vec) store_chunk8<T, true>(W + base, out);
    else
        for (int i = 0; i < n; ++i) W[base + i] = out.v[i];
}

// walks the contiguous chunk range of one work-group through the job table
struct Walker {
    int64_t chunk, end;
    int job;
    __device__ Walker(const DevJob *jobs, int n_jobs, int64_t total) {
        const int64_t g = gridDim.x, w = blockIdx.x;
        chunk = total / g * w + (total % g < w ? total % g : w);
        end = chunk + total / g + (w < total % g ? 1 : 0);
        int lo = 0, hi = n_jobs - 1;                   // last job with chunk_begin <= chunk
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].chunk_begin <= chunk) lo = mid; else hi = mid - 1;
        }
        job = lo;
    }
    __device__ void settle(const DevJob *jobs, int n_jobs) {
        while (job + 1 < n_jobs && jobs[job + 1].chunk_begin <= chunk) ++job;
    }
};

__device__ __forceinline__ void flush_hist(uint32_t *lh, uint32_t *gh) {
    __syncthreads();
    for (int b = threadIdx.x; b < kBins; b += kThreads) {
        const uint32_t c = lh[b];
        if (c) {
            atomicAdd(&gh[b], c);
            lh[b] = 0;
        }
    }
    __syncthreads();
}

#ifndef SCORE_UNROLL
#define SCORE_UNROLL 2
#endif
constexpr int kUnroll = SCORE_UNROLL;   // chunks whose loads are issued back to back (bytes in flight per lane: 4 x 16..72 B)

// Drives one work-group through its chunk range.  on_job(job index, job) -> "this job takes part" is called
// when the walker enters a job; on_elems(job, base, elems) for every 8 elements of every lane.
template <int MODE, bool LOADW, typename JobFn, typename ElemFn>
__device__ __forceinline__ void walk(const DevJob *__restrict__ jobs, int n_jobs, int64_t total, JobFn on_job, ElemFn on_elems) {
    Walker wk(jobs, n_jobs, total);
    int cur_job = -1;
    DevJob j = jobs[wk.job];
    bool active = false;
    int64_t job_end = 0, full_end = 0;
    while (wk.chunk < wk.end) {
        wk.settle(jobs, n_jobs);
        if (wk.job != cur_job) {
            cur_job = wk.job;
            j = jobs[cur_job];
            active = on_job(cur_job, j);
            job_end = cur_job + 1 < n_jobs ? jobs[cur_job + 1].chunk_begin : total;
            full_end = j.chunk_begin + j.numel / kChunk;       // chunks in which every lane has 8 elements
        }
        if (!active) {
            wk.chunk = job_end < wk.end ? job_end : wk.end;
            continue;
        }
        const int64_t base = (wk.chunk - j.chunk_begin) * kChunk + int64_t(threadIdx.x) * 8;
        if (j.vec_ok && wk.chunk + kUnroll <= wk.end && wk.chunk + kUnroll <= full_end) {
            Elems e[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) load_scores<MODE, LOADW>(j, base + int64_t(u) * kChunk, e[u]);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) on_elems(j, base + int64_t(u) * kChunk, e[u]);
            wk.chunk += kUnroll;
        } else {
            Elems e;
            load_scores<MODE, LOADW>(j, base, e);
            on_elems(j, base, e);
            ++wk.chunk;
        }
    }
}

// PROTECT: per-job histograms of the INVERTED key (k-th largest); otherwise per-scope histograms of the
// key after the protection transform.
template <int MODE, bool PROTECT>
__global__ __launch_bounds__(kThreads) void score_hist_kernel(const DevJob *__restrict__ jobs, int n_jobs, int64_t total_chunks,
                                                              const SelState *__restrict__ scope_st, const SelState *__restrict__ prot_st,
                                                              uint32_t *__restrict__ hist, int shift, int bits) {
    __shared__ uint32_t lh[kBins];
    for (int b = threadIdx.x; b < kBins; b += kThreads) lh[b] = 0;
    __syncthreads();
    const int hi_shift = shift + bits;
    const uint32_t bin_mask = (1u << bits) - 1;
    int cur = -1;                                     // histogram the LDS copy belongs to
    SelState ps{};
    bool capped = false;
    float cap_f = 0.f;                                // scores >= cap_f count as FLT_MAX (per-layer protection)
    uint32_t match = 0;
    walk<MODE, MODE != 1>(
        jobs, n_jobs, total_chunks,
        [&](int job, const DevJob &j) {
            ps = prot_st[job];
            if (PROTECT && !ps.active) return false;
            const int hid = PROTECT ? job : j.scope;
            if (hid != cur) {
                if (cur >= 0) flush_hist(lh, hist + size_t(cur) * kBins);
                cur = hid;
            }
            const SelState st = PROTECT ? ps : scope_st[hid];
            // the protection phase selects on the inverted key (k-th largest): its prefix is a prefix of ~key
            const uint32_t P = PROTECT ? ~st.prefix : st.prefix;
            match = hi_shift >= 32 ? 0 : prefix_match(P, hi_shift);
            capped = !PROTECT && ps.active;
            cap_f = key_to_float(~ps.prefix);         // valid once the protection passes ran
            return st.active != 0;
        },
        [&](const DevJob &, int64_t, const Elems &e) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i >= e.n) continue;
                float sc = e.s[i];
                if (capped && sc >= cap_f) sc = 3.4028234663852886e38f;
                const uint32_t b = canonical_bits(sc);
                if (hi_shift >= 32 || (b >> hi_shift) == match) {
                    uint32_t key = key_of_bits(b);
                    if (PROTECT) key = ~key;
                    atomicAdd(&lh[(key >> shift) & bin_mask], 1u);
                }
            }
        });
    if (cur >= 0) flush_hist(lh, hist + size_t(cur) * kBins);
}

// one work-group per histogram: bin holding rank k_rem, refine the prefix, clear the histogram
__global__ __launch_bounds__(kThreads) void score_pick_kernel(SelState *__restrict__ st, uint32_t *__restrict__ hist, int shift) {
    __shared__ uint32_t part[kThreads];
    __shared__ int found_bin;
    __shared__ int64_t found_below;
    SelState s = st[blockIdx.x];
    uint32_t *h = hist + size_t(blockIdx.x) * kBins;
    if (!s.active) return;
    constexpr int per = kBins / kThreads;
    uint32_t c[per];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < per; ++i) {
        c[i] = h[threadIdx.x * per + i];
        h[threadIdx.x * per + i] = 0;
        sum += c[i];
    }
    part[threadIdx.x] = sum;
    if (threadIdx.x == 0) found_bin = -1;
    __syncthreads();
    if (threadIdx.x == 0) {                            // 256 partial sums: a serial scan is a few hundred cycles
        int64_t run = 0;
        for (int t = 0; t < kThreads; ++t) {
            if (run + part[t] >= s.k_rem) {
                found_bin = t;
                found_below = run;
                break;
            }
            run += part[t];
        }
    }
    __syncthreads();
    if (found_bin == int(threadIdx.x)) {
        int64_t run = found_below;
        for (int i = 0; i < per; ++i) {
            if (run + c[i] >= s.k_rem) {
                s.prefix |= uint32_t(threadIdx.x * per + i) << shift;
                s.k_rem -= run;
                st[blockIdx.x] = s;
                break;
            }
            run += c[i];
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void score_apply_kernel(const DevJob *__restrict__ jobs, int n_jobs, int64_t total_chunks,
                                                               const SelState *__restrict__ scope_st, const SelState *__restrict__ prot_st,
                                                               int apply_w) {
    bool capped = false;
    float cap_f = 0.f, thr_f = 0.f;
    walk<MODE, true>(
        jobs, n_jobs, total_chunks,
        [&](int job, const DevJob &j) {
            const SelState ps = prot_st[job];
            capped = ps.active != 0;
            cap_f = key_to_float(~ps.prefix);
            thr_f = key_to_float(scope_st[j.scope].prefix);     // NaN when the threshold is a NaN score: nothing is kept
            return true;
        },
        [&](const DevJob &j, int64_t base, const Elems &e) {
            const bool vec = e.n == 8 && j.vec_ok;
            uint8_t kp[8];
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float sc = e.s[i];
                if (capped && sc >= cap_f) sc = 3.4028234663852886e38f;
                const bool keep = sc > thr_f;                           // masks = score > threshold (:130,146); NaN: false
                kp[i] = keep ? 1 : 0;
                w[i] = keep ? e.w[i] : ieee_mul(e.w[i], 0.f);           // v.data *= mask (:190): a pruned weight keeps its sign
            }
            if (vec) {
                u32x2_t q;
                q.x = kp[0] | (kp[1] << 8) | (kp[2] << 16) | (uint32_t(kp[3]) << 24);
                q.y = kp[4] | (kp[5] << 8) | (kp[6] << 16) | (uint32_t(kp[7]) << 24);
                __builtin_nontemporal_store(q, reinterpret_cast<u32x2_t *>(j.keep + base));
            } else {
                for (int i = 0; i < e.n; ++i) j.keep[base + i] = kp[i];
            }
            if (apply_w && j.W) {
                if (j.dtype == VLMC_BF16) store_w8<bf16_t>(j.W, base, e.n, vec, w);
                else if (j.dtype == VLMC_F16) store_w8<f16_t>(j.W, base, e.n, vec, w);
                else store_w8<f32_t>(j.W, base, e.n, vec, w);
            }
        });
}

// The job table travels as kernel arguments, kTableChunk jobs per (tiny) launch: no host-to-device copy of pageable
// memory, no wait -- SparseGPT's per-block thresholds (1 job) and a global threshold over all 588 linears of a model
// (25 launches) alike stay fully asynchronous.  (Round 1 uploaded tables of more than 4 jobs with hipMemcpyAsync and
// had to wait for the copy because the host buffer was freed on return: the one export that synchronised.)
constexpr int kTableChunk = 24;                    // 24 x (DevJob + 2 SelState) < the 4 KB a launch may carry
struct TableChunk {
    DevJob jobs[kTableChunk];
    SelState scope_st[kTableChunk];
    SelState prot_st[kTableChunk];
};
static_assert(sizeof(TableChunk) <= 3584, "kernel arguments are limited to 4 KB");
__global__ void score_table_kernel(TableChunk t, int first, int n_jobs, int n_scopes, DevJob *jobs, SelState *scope_st,
                                   SelState *prot_st) {
    const int i = threadIdx.x, g = first + i;
    if (i < kTableChunk && g < n_jobs) {
        jobs[g] = t.jobs[i];
        prot_st[g] = t.prot_st[i];
    }
    if (i < kTableChunk && g < n_scopes) scope_st[g] = t.scope_st[i];
}

struct Layout {
    size_t jobs, scope_st, prot_st, hist, total;
};
Layout layout(int n_jobs, int n_scopes) {
    Layout l;
    l.jobs = 0;
    l.scope_st = round_up(size_t(n_jobs) * sizeof(DevJob), 256);
    l.prot_st = l.scope_st + round_up(size_t(n_scopes) * sizeof(SelState), 256);
    l.hist = l.prot_st + round_up(size_t(n_jobs) * sizeof(SelState), 256);
    l.total = l.hist + size_t(n_jobs > n_scopes ? n_jobs : n_scopes) * kBins * sizeof(uint32_t);
    return l;
}

template <int MODE>
int run(const DevJob *dj, int n_jobs, int n_scopes, int64_t total_chunks, SelState *scope_st, SelState *prot_st, uint32_t *hist,
        bool any_protect, int apply_w, hipStream_t st) {
    const int wgs = int(total_chunks < kMaxWgs ? total_chunks : kMaxWgs);
    static const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    if (any_protect)
        for (int p = 0; p < 3; ++p) {
            hipLaunchKernelGGL((score_hist_kernel<MODE, true>), dim3(wgs), dim3(kThreads), 0, st, dj, n_jobs, total_chunks, scope_st,
                               prot_st, hist, shifts[p], widths[p]);
            hipLaunchKernelGGL(score_pick_kernel, dim3(n_jobs), dim3(kThreads), 0, st, prot_st, hist, shifts[p]);
        }
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL((score_hist_kernel<MODE, false>), dim3(wgs), dim3(kThreads), 0, st, dj, n_jobs, total_chunks, scope_st,
                           prot_st, hist, shifts[p], widths[p]);
        hipLaunchKernelGGL(score_pick_kernel, dim3(n_scopes), dim3(kThreads), 0, st, scope_st, hist, shifts[p]);
    }
    hipLaunchKernelGGL((score_apply_kernel<MODE>), dim3(wgs), dim3(kThreads), 0, st, dj, n_jobs, total_chunks, scope_st, prot_st, apply_w);
    VLMC_HIP_CHECK_LAUNCH("vlmc_score_select");
    return VLMC_OK;
}

}  // namespace
}  // namespace vlmc

using namespace vlmc;

extern "C" size_t vlmc_score_select_workspace(int n_jobs, int n_scopes) {
    if (n_jobs <= 0 || n_scopes <= 0) return 0;
    return layout(n_jobs, n_scopes).total;
}

extern "C" int vlmc_score_select(const vlmc_score_job *jobs, int n_jobs, const int64_t *scope_k, int n_scopes, int score_mode,
                                 int apply_weights, void *workspace, size_t workspace_bytes, void *stream) {
    VLMC_REQUIRE(jobs && n_jobs > 0 && scope_k && n_scopes > 0 && n_scopes <= n_jobs, "vlmc_score_select: bad job / scope table");
    VLMC_REQUIRE(score_mode >= VLMC_SCORE_W && score_mode <= VLMC_SCORE_ABSW_S, "vlmc_score_select: bad score mode %d", score_mode);
    const Layout l = layout(n_jobs, n_scopes);
    if (!workspace || workspace_bytes < l.total) {
        set_error("vlmc_score_select: workspace of %zu bytes needed, %zu given", l.total, workspace_bytes);
        return VLMC_EWORKSPACE;
    }
    VLMC_REQUIRE(aligned16(workspace), "vlmc_score_select: workspace must be 16-byte aligned");
    std::vector<char> blob(l.hist, 0);
    DevJob *dj = reinterpret_cast<DevJob *>(blob.data() + l.jobs);
    SelState *sst = reinterpret_cast<SelState *>(blob.data() + l.scope_st), *pst = reinterpret_cast<SelState *>(blob.data() + l.prot_st);
    std::vector<int64_t> scope_numel(n_scopes, 0);
    int64_t chunks = 0;
    bool any_protect = false;
    for (int i = 0; i < n_jobs; ++i) {
        const vlmc_score_job &j = jobs[i];
        VLMC_REQUIRE(j.keep && j.numel > 0, "vlmc_score_select: job %d: null keep mask or empty tensor", i);
        VLMC_REQUIRE(j.W || score_mode == VLMC_SCORE_S, "vlmc_score_select: job %d: this score needs the weights", i);
        VLMC_REQUIRE(j.S || score_mode == VLMC_SCORE_W, "vlmc_score_select: job %d: this score needs S", i);
        VLMC_REQUIRE(j.dtype >= VLMC_F32 && j.dtype <= VLMC_BF16, "vlmc_score_select: job %d: bad dtype %d", i, j.dtype);
        VLMC_REQUIRE(j.scope >= 0 && j.scope < n_scopes, "vlmc_score_select: job %d: scope %d out of range", i, j.scope);
        VLMC_REQUIRE(j.protect_k >= 0 && j.protect_k <= j.numel, "vlmc_score_select: job %d: protect_k out of range", i);
        DevJob &d = dj[i];
        d.W = j.W;
        d.S = j.S;
        d.prev = j.prev_keep;
        d.keep = j.keep;
        d.numel = j.numel;
        d.chunk_begin = chunks;
        d.protect_k = j.protect_k;
        d.scope = j.scope;
        d.dtype = int16_t(j.dtype);
        d.vec_ok = (!j.W || (reinterpret_cast<uintptr_t>(j.W) % 16 == 0)) && (!j.S || aligned16(j.S)) &&
                   (!j.prev_keep || reinterpret_cast<uintptr_t>(j.prev_keep) % 8 == 0) && reinterpret_cast<uintptr_t>(j.keep) % 8 == 0;
        chunks += (j.numel + kChunk - 1) / kChunk;
        scope_numel[j.scope] += j.numel;
        pst[i].prefix = 0;
        pst[i].active = j.protect_k > 0;
        pst[i].k_rem = j.protect_k;
        any_protect |= j.protect_k > 0;
    }
    for (int s = 0; s < n_scopes; ++s) {
        // k == 0 is the reference's `threshold[-1]` on an empty topk: an IndexError there, refused here
        VLMC_REQUIRE(scope_numel[s] > 0 && scope_k[s] >= 1 && scope_k[s] <= scope_numel[s],
                     "vlmc_score_select: scope %d: k=%lld outside [1, %lld]", s, (long long)scope_k[s], (long long)scope_numel[s]);
        sst[s].prefix = 0;
        sst[s].active = 1;
        sst[s].k_rem = scope_k[s];
    }
    hipStream_t st = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    DevJob *ddj = reinterpret_cast<DevJob *>(ws + l.jobs);
    SelState *dss = reinterpret_cast<SelState *>(ws + l.scope_st), *dps = reinterpret_cast<SelState *>(ws + l.prot_st);
    for (int first = 0; first < n_jobs; first += kTableChunk) {
        TableChunk t{};
        for (int i = 0; i < kTableChunk && first + i < n_jobs; ++i) {
            t.jobs[i] = dj[first + i];
            t.prot_st[i] = pst[first + i];
            if (first + i < n_scopes) t.scope_st[i] = sst[first + i];
        }
        hipLaunchKernelGGL(score_table_kernel, dim3(1), dim3(64), 0, st, t, first, n_jobs, n_scopes, ddj, dss, dps);
    }
    if (hipMemsetAsync(ws + l.hist, 0, l.total - l.hist, st) != hipSuccess) {
        set_error("vlmc_score_select: clearing the histograms failed: %s", hipGetErrorString(hipGetLastError()));
        return VLMC_EHIP;
    }
    uint32_t *hist = reinterpret_cast<uint32_t *>(ws + l.hist);
    switch (score_mode) {
        case VLMC_SCORE_W: return run<0>(ddj, n_jobs, n_scopes, chunks, dss, dps, hist, any_protect, apply_weights, st);
        case VLMC_SCORE_S: return run<1>(ddj, n_jobs, n_scopes, chunks, dss, dps, hist, any_protect, apply_weights, st);
        default: return run<2>(ddj, n_jobs, n_scopes, chunks, dss, dps, hist, any_protect, apply_weights, st);
    }
}
