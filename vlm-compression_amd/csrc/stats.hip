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

// VEC consecutive elements per lane as ONE naturally aligned vector load (<= 16 B; 32 B = 2 loads).
template <typename T, int VEC> __device__ __forceinline__ void vec_load(const typename T::raw *p, float *o) {
    using raw = typename T::raw;
    if constexpr (VEC == 1) {
        o[0] = to_f32<T>(p[0]);
    } else {
        struct alignas((sizeof(raw) * VEC > 16) ? 16 : sizeof(raw) * VEC) Pack { raw r[VEC]; };
        const Pack q = *reinterpret_cast<const Pack *>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = to_f32<T>(q.r[i]);
    }
}

template <typename T, int VEC, int UNROLL>
__global__ __launch_bounds__(256) void act_sqnorm_kernel(const typename T::raw *__restrict__ x, int64_t tokens,
                                                         int64_t in_f, int64_t row_stride, int64_t call_stride,
                                                         float *__restrict__ normsq) {
    const int64_t ch = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
    if (ch >= in_f) return;
    const int64_t call = blockIdx.y;
    const typename T::raw *p = x + call * call_stride + ch;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    int64_t t = 0;
    for (; t + UNROLL <= tokens; t += UNROLL) {
        float xv[UNROLL][VEC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vec_load<T, VEC>(p + (t + u) * row_stride, xv[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(xv[u][v], xv[u][v], acc[v]);
    }
    for (; t < tokens; ++t) {
        float xv[VEC];
        vec_load<T, VEC>(p + t * row_stride, xv);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(xv[v], xv[v], acc[v]);
    }
    float *o = normsq + call * in_f + ch;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float r = ieee_sqrt(acc[v]);       // torch.norm(p=2): sqrt of the sum ...
        o[v] = ieee_mul(r, r);             // ... then `** 2` (wanda_pruner.py:81)
    }
}

// s *= float(n/(n+b)); n += b; s += normsq[c] / float(n)   (wanda_pruner.py:77-81)
// One lane per channel walks the calls in order.  The scale factors float(n/(n+b)) (a python
// double division rounded to fp32) are the same for every channel: they are computed once per
// workgroup into LDS, off the dependent chain, which is then one multiply and one add per call;
// the per-call divisions normsq/float(n) do not depend on the accumulator and pipeline freely.
constexpr int kUpdChannels = 64;    // channels per workgroup
constexpr int kUpdGroups = 4;       // call groups (256 threads = 64 channels x 4 groups)
constexpr int kUpdChunk = 128;      // calls staged in LDS at a time (32 KB)
__global__ __launch_bounds__(256) void scaler_update_kernel(float *__restrict__ s, int64_t in_f, int64_t n0,
                                                            const float *__restrict__ normsq, int64_t n_calls,
                                                            int64_t batch, float *__restrict__ sqrt_out) {
    // The normsq loads are the latency problem of this tiny kernel (calls x in fp32, read once):
    // 256 threads fetch and pre-divide a [128 calls x 64 channels] panel into LDS with all loads in
    // flight at once, then 64 lanes run the short dependent chain (one mul + one add per call).
    __shared__ float q[kUpdChunk][kUpdChannels];
    __shared__ float fac[kUpdChunk];
    const int chl = threadIdx.x % kUpdChannels, grp = threadIdx.x / kUpdChannels;
    const int64_t ch = int64_t(blockIdx.x) * kUpdChannels + chl;
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
            v[i] = (live && c < cn) ? normsq[(c0 + c) * in_f + ch] : 0.f;
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

template <typename T>
static int launch_sqnorm(const void *x, int64_t n_calls, int64_t tokens, int64_t in_f, int64_t row_stride,
                         int64_t call_stride, float *normsq, hipStream_t st) {
    using raw = typename T::raw;
    const raw *xp = static_cast<const raw *>(x);
    const size_t esz = sizeof(raw);
    // widest vector such that every lane's load is naturally aligned
    auto ok = [&](int vec) {
        const size_t bytes = vec * esz;
        return in_f % vec == 0 && (reinterpret_cast<uintptr_t>(x) % bytes) == 0 && (row_stride * esz) % bytes == 0 &&
               (call_stride * esz) % bytes == 0;
    };
    // Widest legal vector that still yields enough lanes to fill 256 CUs; small problems fall
    // back to the narrowest legal vector (4-byte lanes at least) to get more waves in flight.
    const int64_t want_threads = 256 * 256 * 2;
    const int max_vec = esz == 2 ? 8 : 4;
    int vec = 1, narrowest = 0;
    bool chosen = false;
    for (int v = max_vec; v >= 2 && !chosen; v >>= 1) {
        if (!ok(v)) continue;
        narrowest = v;
        if (n_calls * (in_f / v) >= want_threads) { vec = v; chosen = true; }
    }
    if (!chosen && narrowest) vec = narrowest;
    if (const char *e = getenv("VLMC_SQNORM_VEC")) {   // tuning override (must be legal)
        const int v = atoi(e);
        if ((v == 1 || v == 2 || v == 4 || (v == 8 && esz == 2)) && (v == 1 || ok(v))) vec = v;
    }
    const int threads = 256;
    auto grid_for = [&](int v) { return dim3(unsigned((in_f / v + threads - 1) / threads), unsigned(n_calls)); };
#define VLMC_LAUNCH_SQ(V, U)                                                                                     \
    hipLaunchKernelGGL((act_sqnorm_kernel<T, V, U>), grid_for(V), dim3(threads), 0, st, xp, tokens, in_f, row_stride, \
                       call_stride, normsq)
    switch (vec) {
        case 8:
            if constexpr (sizeof(raw) == 2) VLMC_LAUNCH_SQ(8, 8);
            break;
        case 4: VLMC_LAUNCH_SQ(4, 8); break;
        case 2: VLMC_LAUNCH_SQ(2, 16); break;
        default: VLMC_LAUNCH_SQ(1, 16); break;
    }
#undef VLMC_LAUNCH_SQ
    VLMC_HIP_CHECK_LAUNCH("vlmc_act_sqnorm");
    return VLMC_OK;
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_act_sqnorm(const void *x, int dtype, int64_t n_calls, int64_t tokens, int64_t in_features,
                               int64_t row_stride, int64_t call_stride, float *normsq, void *stream) {
    VLMC_REQUIRE(x && normsq, "vlmc_act_sqnorm: null pointer");
    VLMC_REQUIRE(n_calls >= 0 && tokens >= 0 && in_features > 0, "vlmc_act_sqnorm: bad shape calls=%lld tokens=%lld in=%lld",
                 (long long)n_calls, (long long)tokens, (long long)in_features);
    VLMC_REQUIRE(row_stride >= in_features, "vlmc_act_sqnorm: row_stride %lld < in_features %lld", (long long)row_stride,
                 (long long)in_features);
    VLMC_REQUIRE(n_calls <= 65535, "vlmc_act_sqnorm: n_calls %lld > 65535", (long long)n_calls);
    if (n_calls == 0) return VLMC_OK;
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case VLMC_F32: return launch_sqnorm<f32_t>(x, n_calls, tokens, in_features, row_stride, call_stride, normsq, st);
        case VLMC_F16: return launch_sqnorm<f16_t>(x, n_calls, tokens, in_features, row_stride, call_stride, normsq, st);
        case VLMC_BF16: return launch_sqnorm<bf16_t>(x, n_calls, tokens, in_features, row_stride, call_stride, normsq, st);
    }
    set_error("vlmc_act_sqnorm: unknown dtype %d", dtype);
    return VLMC_EINVAL;
}

extern "C" int vlmc_wanda_scaler_update(float *scaler_row, int64_t in_features, int64_t nsamples_before,
                                        const float *normsq, int64_t n_calls, int64_t batch, float *sqrt_out,
                                        void *stream) {
    VLMC_REQUIRE(scaler_row && (normsq || n_calls == 0), "vlmc_wanda_scaler_update: null pointer");
    VLMC_REQUIRE(in_features > 0 && n_calls >= 0 && batch > 0 && nsamples_before >= 0,
                 "vlmc_wanda_scaler_update: bad arguments in=%lld calls=%lld batch=%lld n0=%lld", (long long)in_features,
                 (long long)n_calls, (long long)batch, (long long)nsamples_before);
    if (n_calls == 0 && !sqrt_out) return VLMC_OK;
    hipLaunchKernelGGL(scaler_update_kernel, dim3(unsigned((in_features + kUpdChannels - 1) / kUpdChannels)), dim3(256), 0,
                       as_stream(stream), scaler_row, in_features, nsamples_before, normsq, n_calls, batch, sqrt_out);
    VLMC_HIP_CHECK_LAUNCH("vlmc_wanda_scaler_update");
    return VLMC_OK;
}
