// K1: Wanda activation statistics (replaces WrappedGPT.add_batch,
// /root/reference/lavis/compression/pruners/wanda_pruner.py:68-81).
//
// HBM-bound streaming reduction over tokens.  Layout: activations [calls, tokens, in]
// (in contiguous).  One lane owns VEC consecutive channels of ONE call and walks the
// tokens in order, so the fp32 result is the same sequential fma chain torch's CPU
// norm kernel performs; lanes of a wave cover 64*VEC consecutive channels, so every
// wave-instruction reads one contiguous 64*VEC*sizeof(T) segment of a token row.
// Algorithmic bytes: calls*tokens*in*sizeof(T) read + calls*in*4 written.
#include <cstdlib>

#include "common.hpp"

namespace vlmc {

// VEC consecutive elements per lane as ONE naturally aligned vector load (<= 16 B), non-temporal: the
// activations are read exactly once, so they should not displace the weights in L2 / Infinity Cache.
template <int BYTES> struct NtWord;
template <> struct NtWord<2> { using type = uint16_t; };
template <> struct NtWord<4> { using type = uint32_t; };
template <> struct NtWord<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct NtWord<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <typename T, int VEC> __device__ __forceinline__ void vec_load(const typename T::raw *p, float *o) {
    using raw = typename T::raw;
    using word = typename NtWord<sizeof(raw) * VEC>::type;
    const word w = __builtin_nontemporal_load(reinterpret_cast<const word *>(p));
    raw r[VEC];
    __builtin_memcpy(r, &w, sizeof(w));
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = to_f32<T>(r[i]);
}

// One hook input of a batched launch (all jobs share dtype and the number of calls).
struct SqJob {
    const void *x;
    float *out;                 // [n_calls, out_stride] (first in_f columns written)
    const int32_t *call_tokens; // device [n_calls] or null: token rows of call c that count (a padded group of ragged samples)
    int64_t row_stride, call_stride, out_stride;
    int32_t in_f, tokens, vec, wg_end;   // wg_end: exclusive prefix end of this job's workgroups in grid.x
};
constexpr int kMaxStatJobs = 12;
struct SqBatch {
    SqJob job[kMaxStatJobs];
    int32_t n;
};

template <typename T, int VEC, int UNROLL>
__device__ __forceinline__ void sqnorm_chain(const typename T::raw *__restrict__ p, int tokens, int64_t row_stride,
                                             float *__restrict__ o) {
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    int t = 0;
    for (; t + UNROLL <= tokens; t += UNROLL) {
        float xv[UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vec_load<T, VEC>(p + int64_t(t + u) * row_stride, xv[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(xv[u][v], xv[u][v], acc[v]);
    }
    if (t < tokens) {
        // tail: the same UNROLL loads in flight (rows past the end re-read the last row), chain over the
        // valid rows only -- short inputs (16 decoder tokens) are one such block
        float xv[UNROLL][VEC];
        const int left = tokens - t;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vec_load<T, VEC>(p + int64_t(t + (u < left ? u : left - 1)) * row_stride, xv[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = u < left ? __builtin_fmaf(xv[u][v], xv[u][v], acc[v]) : acc[v];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float r = ieee_sqrt(acc[v]);       // torch.norm(p=2): sqrt of the sum ...
        o[v] = ieee_mul(r, r);             // ... then `** 2` (wanda_pruner.py:81)
    }
}

// 1-D grid, job-major: job j owns tiles_j x n_calls single-wave workgroups (tile fastest), one lane = `vec`
// channels of one call.  Job-major order keeps the chip on one activation tensor at a time (DRAM locality of
// back-to-back launches) without their launch gaps.  Every lane keeps UNROLL*VEC = 64 elements in flight;
// with calls x channels lanes that covers the HBM latency-bandwidth product of this streaming reduction.
template <typename T>
__global__ __launch_bounds__(64) void act_sqnorm_kernel(const SqBatch b) {
    int j = 0;
    while (j + 1 < b.n && int(blockIdx.x) >= b.job[j].wg_end) ++j;
    const SqJob &jb = b.job[j];
    const int local = int(blockIdx.x) - (j ? b.job[j - 1].wg_end : 0);
    const int vec = jb.vec;
    const int tiles = (jb.in_f / vec + 63) / 64;
    const int call = local / tiles, wg = local - call * tiles;
    const int64_t ch = (int64_t(wg) * 64 + threadIdx.x) * vec;
    if (ch >= jb.in_f) return;
    const typename T::raw *p = static_cast<const typename T::raw *>(jb.x) + call * jb.call_stride + ch;
    float *o = jb.out + call * jb.out_stride + ch;
    // (a call's own token count: the rows behind it are padding -- same chain over the same rows as the unpadded call)
    const int tokens = jb.call_tokens ? min(jb.call_tokens[call], jb.tokens) : jb.tokens;
    if constexpr (sizeof(typename T::raw) == 2) {
        if (vec == 8) { sqnorm_chain<T, 8, 8>(p, tokens, jb.row_stride, o); return; }
    }
    if (vec == 4) sqnorm_chain<T, 4, 16>(p, tokens, jb.row_stride, o);
    else if (vec == 2) sqnorm_chain<T, 2, 32>(p, tokens, jb.row_stride, o);
    else sqnorm_chain<T, 1, 32>(p, tokens, jb.row_stride, o);
}

// s *= float(n/(n+b)); n += b; s += normsq[c] / float(n)   (wanda_pruner.py:77-81)
// One lane per channel walks the calls in order.  The scale factors float(n/(n+b)) (a python
// double division rounded to fp32) are the same for every channel: they are computed once per
// workgroup into LDS, off the dependent chain, which is then one multiply and one add per call;
// the per-call divisions normsq/float(n) do not depend on the accumulator and pipeline freely.
constexpr int kUpdChannels = 64;    // channels per workgroup
constexpr int kUpdGroups = 4;       // call groups (256 threads = 64 channels x 4 groups)
constexpr int kUpdChunk = 128;      // calls staged in LDS at a time (32 KB)
struct UpdJob {
    float *s;                   // [in_f] running mean, in/out
    const float *normsq;        // [n_calls, nsq_stride]
    float *sqrt_out;            // [in_f] or null
    int64_t nsq_stride;
    int32_t in_f, wg_end;
};
struct UpdBatch {
    UpdJob job[kMaxStatJobs];
    int32_t n;
};
__global__ __launch_bounds__(256) void scaler_update_kernel(const UpdBatch b, int64_t n0, int64_t n_calls, int64_t batch) {
    int jj = 0;
    while (jj + 1 < b.n && int(blockIdx.x) >= b.job[jj].wg_end) ++jj;
    const UpdJob &jb = b.job[jj];
    const int wg = int(blockIdx.x) - (jj ? b.job[jj - 1].wg_end : 0);
    float *__restrict__ s = jb.s;
    const float *__restrict__ normsq = jb.normsq;
    float *__restrict__ sqrt_out = jb.sqrt_out;
    const int64_t in_f = jb.in_f, nstride = jb.nsq_stride;
    // The normsq loads are the latency problem of this tiny kernel (calls x in fp32, read once):
    // 256 threads fetch and pre-divide a [128 calls x 64 channels] panel into LDS with all loads in
    // flight at once, then 64 lanes run the short dependent chain (one mul + one add per call).
    __shared__ float q[kUpdChunk][kUpdChannels];
    __shared__ float fac[kUpdChunk];
    const int chl = threadIdx.x % kUpdChannels, grp = threadIdx.x / kUpdChannels;
    const int64_t ch = int64_t(wg) * kUpdChannels + chl;
    const bool live = ch < in_f;
    // a fresh statistic starts from zeros (WrappedGPT.__init__, wanda_pruner.py:62)
    float acc = (live && !(n0 == 0 && n_calls > 0)) ? s[ch] : 0.f;
    for (int64_t c0 = 0; c0 < n_calls; c0 += kUpdChunk) {
        const int cn = int((n_calls - c0 < kUpdChunk) ? (n_calls - c0) : kUpdChunk);
        __syncthreads();
        if (int(threadIdx.x) < cn) {
            const int64_t n = n0 + (c0 + threadIdx.x) * batch;
            fac[threadIdx.x] = float(double(n) / double(n + batch));   // python float -> fp32 scalar
        }
        float v[kUpdChunk / kUpdGroups];
#pragma unroll
        for (int i = 0; i < kUpdChunk / kUpdGroups; ++i) {
            const int c = i * kUpdGroups + grp;
            v[i] = (live && c < cn) ? normsq[(c0 + c) * nstride + ch] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < kUpdChunk / kUpdGroups; ++i) {
            const int c = i * kUpdGroups + grp;
            if (c < cn) q[c][chl] = ieee_div(v[i], float(n0 + (c0 + c + 1) * batch));
        }
        __syncthreads();
        if (grp == 0) {
            for (int c = 0; c < cn; ++c) acc = ieee_add(ieee_mul(acc, fac[c]), q[c][chl]);
        }
    }
    if (live && grp == 0) {
        s[ch] = acc;
        if (sqrt_out) sqrt_out[ch] = ieee_sqrt(acc);   // torch.sqrt(scaler_row), wanda_pruner.py:318
    }
}

// widest vector such that every lane's load is naturally aligned, but still enough lanes to fill 256 CUs
static int pick_vec(const vlmc_stat_job &j, size_t esz, int64_t n_calls) {
    auto ok = [&](int vec) {
        const size_t bytes = vec * esz;
        return j.in_features % vec == 0 && (reinterpret_cast<uintptr_t>(j.x) % bytes) == 0 &&
               (j.row_stride * esz) % bytes == 0 && (j.call_stride * esz) % bytes == 0;
    };
    const int64_t want_threads = 256 * 256 * 2;
    const int max_vec = esz == 2 ? 8 : 4;
    int vec = 1, narrowest = 0;
    bool chosen = false;
    for (int v = max_vec; v >= 2 && !chosen; v >>= 1) {
        if (!ok(v)) continue;
        narrowest = v;
        if (n_calls * (j.in_features / v) >= want_threads) { vec = v; chosen = true; }
    }
    if (!chosen && narrowest) vec = narrowest;
    if (const char *e = getenv("VLMC_SQNORM_VEC")) {   // tuning override (must be legal)
        const int v = atoi(e);
        if ((v == 1 || v == 2 || v == 4 || (v == 8 && esz == 2)) && (v == 1 || ok(v))) vec = v;
    }
    return vec;
}

template <typename T>
static int launch_sqnorm(const vlmc_stat_job *jobs, int n_jobs, int64_t n_calls, hipStream_t st) {
    for (int base = 0; base < n_jobs; base += kMaxStatJobs) {
        SqBatch b;
        b.n = (n_jobs - base < kMaxStatJobs) ? n_jobs - base : kMaxStatJobs;
        int64_t wgs = 0;
        for (int i = 0; i < b.n; ++i) {
            const vlmc_stat_job &j = jobs[base + i];
            SqJob &d = b.job[i];
            d.x = j.x; d.out = j.normsq; d.call_tokens = j.call_tokens;
            d.row_stride = j.row_stride; d.call_stride = j.call_stride; d.out_stride = j.normsq_stride;
            d.in_f = int32_t(j.in_features); d.tokens = int32_t(j.tokens);
            d.vec = pick_vec(j, sizeof(typename T::raw), n_calls);
            wgs += int64_t((j.in_features / d.vec + 63) / 64) * n_calls;
            d.wg_end = int32_t(wgs);
        }
        if (wgs >= (int64_t(1) << 31)) {
            set_error("vlmc_act_sqnorm: grid too large");
            return VLMC_EINVAL;
        }
        VLMC_LAUNCH_TIMED((act_sqnorm_kernel<T>), dim3(unsigned(wgs)), dim3(64), st, b);
    }
    VLMC_HIP_CHECK_LAUNCH("vlmc_act_sqnorm");
    return VLMC_OK;
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_act_sqnorm_batch(const vlmc_stat_job *jobs, int n_jobs, int dtype, int64_t n_calls, void *stream) {
    VLMC_REQUIRE(jobs && n_jobs > 0, "vlmc_act_sqnorm_batch: no jobs");
    VLMC_REQUIRE(n_calls >= 0 && n_calls <= 65535, "vlmc_act_sqnorm_batch: n_calls %lld out of range", (long long)n_calls);
    for (int i = 0; i < n_jobs; ++i) {
        const vlmc_stat_job &j = jobs[i];
        VLMC_REQUIRE(j.x && j.normsq, "vlmc_act_sqnorm: null pointer (job %d)", i);
        VLMC_REQUIRE(j.tokens >= 0 && j.tokens < (1ll << 31) && j.in_features > 0 && j.in_features < (1ll << 31),
                     "vlmc_act_sqnorm: bad shape calls=%lld tokens=%lld in=%lld", (long long)n_calls, (long long)j.tokens,
                     (long long)j.in_features);
        VLMC_REQUIRE(j.row_stride >= j.in_features, "vlmc_act_sqnorm: row_stride %lld < in_features %lld",
                     (long long)j.row_stride, (long long)j.in_features);
        VLMC_REQUIRE(j.normsq_stride >= j.in_features, "vlmc_act_sqnorm: normsq_stride %lld < in_features %lld",
                     (long long)j.normsq_stride, (long long)j.in_features);
    }
    if (n_calls == 0) return VLMC_OK;
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case VLMC_F32: return launch_sqnorm<f32_t>(jobs, n_jobs, n_calls, st);
        case VLMC_F16: return launch_sqnorm<f16_t>(jobs, n_jobs, n_calls, st);
        case VLMC_BF16: return launch_sqnorm<bf16_t>(jobs, n_jobs, n_calls, st);
    }
    set_error("vlmc_act_sqnorm: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}

extern "C" int vlmc_act_sqnorm(const void *x, int dtype, int64_t n_calls, int64_t tokens, int64_t in_features,
                               int64_t row_stride, int64_t call_stride, float *normsq, void *stream) {
    const vlmc_stat_job j{x, normsq, in_features, tokens, row_stride, call_stride, in_features, nullptr};
    return vlmc_act_sqnorm_batch(&j, 1, dtype, n_calls, stream);
}

extern "C" int vlmc_wanda_scaler_update_batch(const vlmc_update_job *jobs, int n_jobs, int64_t nsamples_before,
                                              int64_t n_calls, int64_t batch, void *stream) {
    VLMC_REQUIRE(jobs && n_jobs > 0, "vlmc_wanda_scaler_update_batch: no jobs");
    VLMC_REQUIRE(n_calls >= 0 && batch > 0 && nsamples_before >= 0,
                 "vlmc_wanda_scaler_update: bad arguments calls=%lld batch=%lld n0=%lld", (long long)n_calls,
                 (long long)batch, (long long)nsamples_before);
    bool any_sqrt = false;
    for (int i = 0; i < n_jobs; ++i) {
        const vlmc_update_job &j = jobs[i];
        VLMC_REQUIRE(j.scaler_row && (j.normsq || n_calls == 0), "vlmc_wanda_scaler_update: null pointer (job %d)", i);
        VLMC_REQUIRE(j.in_features > 0 && j.in_features < (1ll << 31) && (n_calls == 0 || j.normsq_stride >= j.in_features),
                     "vlmc_wanda_scaler_update: bad arguments in=%lld stride=%lld", (long long)j.in_features,
                     (long long)j.normsq_stride);
        any_sqrt = any_sqrt || j.sqrt_out;
    }
    if (n_calls == 0 && !any_sqrt) return VLMC_OK;
    for (int base = 0; base < n_jobs; base += kMaxStatJobs) {
        UpdBatch b;
        b.n = (n_jobs - base < kMaxStatJobs) ? n_jobs - base : kMaxStatJobs;
        int wgs = 0;
        for (int i = 0; i < b.n; ++i) {
            const vlmc_update_job &j = jobs[base + i];
            UpdJob &d = b.job[i];
            d.s = j.scaler_row; d.normsq = j.normsq; d.sqrt_out = j.sqrt_out; d.nsq_stride = j.normsq_stride;
            d.in_f = int32_t(j.in_features);
            wgs += int((j.in_features + kUpdChannels - 1) / kUpdChannels);
            d.wg_end = wgs;
        }
        hipLaunchKernelGGL(scaler_update_kernel, dim3(unsigned(wgs)), dim3(256), 0, as_stream(stream), b, nsamples_before,
                           n_calls, batch);
    }
    VLMC_HIP_CHECK_LAUNCH("vlmc_wanda_scaler_update");
    return VLMC_OK;
}

extern "C" int vlmc_wanda_scaler_update(float *scaler_row, int64_t in_features, int64_t nsamples_before,
                                        const float *normsq, int64_t n_calls, int64_t batch, float *sqrt_out,
                                        void *stream) {
    const vlmc_update_job j{scaler_row, normsq, sqrt_out, in_features, in_features};
    return vlmc_wanda_scaler_update_batch(&j, 1, nsamples_before, n_calls, batch, stream);
}
