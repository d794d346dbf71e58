// vlmc_row_mean: batch-invariant mean over the last dimension of an fp32 matrix.
//
// The norms of the language models square their input in fp32 and average over the hidden dimension --
// `hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)` (transformers' T5LayerNorm, called from modeling_t5.py's
// blocks; LlamaRMSNorm likewise).  torch's reduction kernel picks its launch configuration -- how many threads and blocks share
// one output -- by the NUMBER of outputs: 4 rows (one calibration sample with a 4-token answer) are summed in another order
// than the same 4 rows inside a group of 512, and the last bit of the mean differs (measured: 72 of 512 rows of a Flan-T5-XL
// decoder block's first norm).  The reference replays every block one sample per forward (wanda_pruner.py:308-311, :343-346);
// the grouped replay must give a sample the bits its own forward would, so during a replay this reduction runs here: one
// wave per row, lane l adds elements 4 l + 256 i .. + 3 in ascending i, the 64 partial sums meet in a fixed butterfly, one
// IEEE division by n.  A row's mean depends on the row and n only.  HBM-bound (4 B read per element), one launch.
#include "common.hpp"
#include "mfma.hpp"
#include "softmax_order.hpp"

#include <type_traits>

namespace vlmc {
namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

}  // namespace

__global__ __launch_bounds__(256) void row_mean_kernel(const float *__restrict__ x, int64_t rows, int n, int64_t ldx, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = x + row * ldx;
    float acc = 0.f;
    const bool vec = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    for (int c0 = 4 * lane; c0 < n; c0 += 256) {
        float v[4];
        if (vec && c0 + 3 < n) {
            const f32x4v q = *reinterpret_cast<const f32x4v *>(p + c0);
            v[0] = q[0], v[1] = q[1], v[2] = q[2], v[3] = q[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = c0 + j < n ? p[c0 + j] : 0.f;
        }
        acc = ieee_add(ieee_add(ieee_add(ieee_add(acc, v[0]), v[1]), v[2]), v[3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = ieee_add(acc, __shfl_xor(acc, off, kWave));
    if (lane == 0) out[row] = ieee_div(acc, float(n));
}


// ---- vlmc_softmax_rows: padding-invariant softmax over the last dimension (include/vlmc.h) --------------------------------------
// The canonical order (softmax_order.hpp; the fused attention kernel forms the same sums from its accumulators): element j
// belongs to class j mod 16; 16 lanes per row, lane c walks its class in ascending j.  Three passes over the row (L1 / L2).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const typename TI::raw *__restrict__ x, int64_t rows, int n, int64_t ldx,
                                                           typename TO::raw *__restrict__ y, int64_t ldy) {
    const int c = threadIdx.x & 15;
    const int64_t row = int64_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;                                               // (whole 16-lane groups: the shuffles below stay inside a group)
    const typename TI::raw *p = x + row * ldx;
    typename TO::raw *q = y + row * ldy;
    float mx = -__builtin_inff();
    for (int j = c; j < n; j += 16) mx = fmaxf(mx, to_f32<TI>(p[j]));
    mx = softmax_class_max(mx);
    float sum = 0.f;
    for (int j = c; j < n; j += 16) sum = ieee_add(sum, softmax_exp(to_f32<TI>(p[j]), mx));
    sum = softmax_class_tree(sum);
    const float inv = softmax_inv(sum);
    for (int j = c; j < n; j += 16) {
        const float r = softmax_prob(softmax_exp(to_f32<TI>(p[j]), mx), inv);
        if constexpr (std::is_same<TO, f32_t>::value) q[j] = r;
        else q[j] = from_f32<TO>(r);
    }
}

// ---- vlmc_rms_norm: the whole RMS norm of a language-model block in one pass -------------------------------------------------
//   y = w * wd(x * rsqrt(mean(float(x)^2) + eps))          (transformers' T5LayerNorm.forward / LlamaRMSNorm.forward, op for op)
// The model files spell it as seven launches -- to(float32), pow(2), mean(-1), + eps, rsqrt, x * r (an fp32 [rows, n] product),
// to(dtype), weight * h -- ~500 MB of traffic for a 33 MB activation, and on one rank's share of the calibration set seven
// dispatches of host time.  Here: one wave per row sums the squares in EXACTLY the order of row_mean_kernel above (lane l:
// elements 4 l + 256 i .. + 3 in ascending i, the same butterfly, the same division), then reads the row again (L2) and writes
// y -- every intermediate rounded where the op sequence rounds it.  `rsqrt_mode` picks how 1 / sqrt is formed -- 0: in double,
// rounded to float: what torch.rsqrt(float) IS on this stack (ATen calls `::rsqrt(a)`, which HIP resolves to the double overload:
// tools/rsqrt_probe.py, 0 of 2.36 M values differ; v_rsq_f32 differs in 11 %); 1: v_rsq_f32; 2: IEEE fp32 1 / sqrt -- the caller
// establishes which one is torch's ONCE (vlmc/forward.py: against torch.rsqrt itself, then against the module's own forward; the
// fused norm is only installed for modules whose own forward it reproduces exactly).
template <typename T>
__global__ __launch_bounds__(256) void rms_norm_kernel(const uint16_t *__restrict__ x, int64_t rows, int n, int64_t ldx,
                                                       const uint16_t *__restrict__ w, float eps, int rsqrt_mode,
                                                       uint16_t *__restrict__ y, int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t *p = x + row * ldx;
    uint16_t *q = y + row * ldy;
    const bool vec = (ldx & 3) == 0 && (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0 && (reinterpret_cast<uintptr_t>(y) & 7u) == 0 &&
                     (reinterpret_cast<uintptr_t>(w) & 7u) == 0;
    auto load4 = [&](const uint16_t *base, int c0, uint16_t (&e)[4]) {
        if (vec && c0 + 3 < n) {
            const uint2 v = *reinterpret_cast<const uint2 *>(base + c0);
            e[0] = uint16_t(v.x & 0xffffu), e[1] = uint16_t(v.x >> 16), e[2] = uint16_t(v.y & 0xffffu), e[3] = uint16_t(v.y >> 16);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = c0 + j < n ? base[c0 + j] : uint16_t(0);
        }
    };
    float acc = 0.f;
    for (int c0 = 4 * lane; c0 < n; c0 += 256) {
        uint16_t e[4];
        load4(p, c0, e);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float f = to_f32<T>(e[j]);
            v[j] = ieee_mul(f, f);                                             // float(x).pow(2)
        }
        acc = ieee_add(ieee_add(ieee_add(ieee_add(acc, v[0]), v[1]), v[2]), v[3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = ieee_add(acc, __shfl_xor(acc, off, kWave));
    const float var = ieee_add(ieee_div(acc, float(n)), eps);                   // mean(-1) + eps   (every lane holds the same sum)
    float r;
    if (rsqrt_mode == 0) r = float(1.0 / __builtin_sqrt(double(var)));         // ::rsqrt(float) resolves to the double overload in ATen's HIP build
    else if (rsqrt_mode == 1) r = __builtin_amdgcn_rsqf(var);
    else r = ieee_div(1.0f, __builtin_sqrtf(var));
    for (int c0 = 4 * lane; c0 < n; c0 += 256) {
        uint16_t e[4], g[4], o[4];
        load4(p, c0, e);
        load4(w, c0, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // The fp32 product is ROUNDED to fp32 before it is rounded to the dtype, as two torch kernels do.  Left to itself the
            // compiler folds multiply + convert into v_fma_mixlo_f16, which rounds the exact product once: x r = 0.668701171875
            // (an fp32 value that is an exact tie between two fp16 neighbours) then goes the other way (found by the bitwise
            // self-check of vlmc/forward.py on fp16 norms).  The empty asm keeps the product a value of its own.
            float xr = ieee_mul(to_f32<T>(e[j]), r);
            asm volatile("" : "+v"(xr));
            const uint16_t h = from_f32<T>(xr);                              // (x * rsqrt(..)).to(dtype)
            float wh = ieee_mul(to_f32<T>(g[j]), to_f32<T>(h));
            asm volatile("" : "+v"(wh));
            o[j] = from_f32<T>(wh);                                          // weight * h, rounded once
        }
        if (vec && c0 + 3 < n) {
            *reinterpret_cast<uint2 *>(q + c0) = uint2{uint32_t(o[0]) | uint32_t(o[1]) << 16, uint32_t(o[2]) | uint32_t(o[3]) << 16};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < n) q[c0 + j] = o[j];
        }
    }
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_rms_norm(const void *x, int dtype, int64_t rows, int64_t n, int64_t ldx, const void *weight, float eps, int rsqrt_mode,
                             void *y, int64_t ldy, void *stream) {
    VLMC_REQUIRE(dtype == VLMC_F16 || dtype == VLMC_BF16, "vlmc_rms_norm: dtype must be VLMC_F16 or VLMC_BF16");
    VLMC_REQUIRE(x && weight && y, "vlmc_rms_norm: null pointer");
    VLMC_REQUIRE(rows >= 0 && n > 0 && n < (int64_t(1) << 30) && ldx >= n && ldy >= n && rows < (int64_t(1) << 32), "vlmc_rms_norm: bad shape");
    VLMC_REQUIRE(rsqrt_mode >= 0 && rsqrt_mode <= 2, "vlmc_rms_norm: rsqrt_mode 0 (double), 1 (v_rsq_f32) or 2 (fp32 1 / sqrt)");
    if (rows == 0) return VLMC_OK;
    const dim3 grid{unsigned((rows + 3) / 4)}, block{256};
    hipStream_t s = as_stream(stream);
    if (dtype == VLMC_F16)
        hipLaunchKernelGGL(rms_norm_kernel<f16_t>, grid, block, 0, s, static_cast<const uint16_t *>(x), rows, int(n), ldx,
                           static_cast<const uint16_t *>(weight), eps, rsqrt_mode, static_cast<uint16_t *>(y), ldy);
    else
        hipLaunchKernelGGL(rms_norm_kernel<bf16_t>, grid, block, 0, s, static_cast<const uint16_t *>(x), rows, int(n), ldx,
                           static_cast<const uint16_t *>(weight), eps, rsqrt_mode, static_cast<uint16_t *>(y), ldy);
    VLMC_HIP_CHECK_LAUNCH("vlmc_rms_norm");
    return VLMC_OK;
}

extern "C" int vlmc_row_mean(const float *x, int64_t rows, int64_t n, int64_t ldx, float *out, void *stream) {
    VLMC_REQUIRE(x && out, "vlmc_row_mean: null pointer");
    VLMC_REQUIRE(rows >= 0 && n > 0 && n < (int64_t(1) << 30) && ldx >= n && rows < (int64_t(1) << 32), "vlmc_row_mean: bad shape");
    if (rows == 0) return VLMC_OK;
    hipLaunchKernelGGL(row_mean_kernel, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, rows, int(n), ldx, out);
    VLMC_HIP_CHECK_LAUNCH("vlmc_row_mean");
    return VLMC_OK;
}

extern "C" int vlmc_softmax_rows(const void *x, int in_dtype, int64_t rows, int64_t n, int64_t ldx, void *y, int out_dtype, int64_t ldy,
                                 void *stream) {
    VLMC_REQUIRE(x && y, "vlmc_softmax_rows: null pointer");
    VLMC_REQUIRE(rows >= 0 && n > 0 && n < (int64_t(1) << 24) && ldx >= n && ldy >= n && rows < (int64_t(1) << 33), "vlmc_softmax_rows: bad shape");
    VLMC_REQUIRE((in_dtype == VLMC_F32 && out_dtype == VLMC_F32) || ((in_dtype == VLMC_F16 || in_dtype == VLMC_BF16) &&
                 (out_dtype == in_dtype || out_dtype == VLMC_F32)),
                 "vlmc_softmax_rows: fp32 -> fp32, or fp16 / bf16 -> the same dtype or fp32");
    if (rows == 0) return VLMC_OK;
    const dim3 grid{unsigned((rows + 15) / 16)}, block{256};
    hipStream_t s = as_stream(stream);
#define VLMC_SM(TI, TO)                                                                                                          \
    hipLaunchKernelGGL((softmax_rows_kernel<TI, TO>), grid, block, 0, s, static_cast<const typename TI::raw *>(x), rows, int(n), ldx,   \
                       static_cast<typename TO::raw *>(y), ldy)
    if (in_dtype == VLMC_F32) VLMC_SM(f32_t, f32_t);
    else if (in_dtype == VLMC_F16 && out_dtype == VLMC_F16) VLMC_SM(f16_t, f16_t);
    else if (in_dtype == VLMC_F16) VLMC_SM(f16_t, f32_t);
    else if (out_dtype == VLMC_BF16) VLMC_SM(bf16_t, bf16_t);
    else VLMC_SM(bf16_t, f32_t);
#undef VLMC_SM
    VLMC_HIP_CHECK_LAUNCH("vlmc_softmax_rows");
    return VLMC_OK;
}
