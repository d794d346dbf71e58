// vlmc_gelu: the activation of a replayed block's feed-forward with ONE instruction sequence for every element.
//
// `self.act(self.fc1(x))` (eva_vit.py:62-64, nn.GELU()), the gated GELU of T5 v1.1 (modeling_t5.py:337-346) run through torch's
// elementwise kernels, and those compute a tensor's LAST partial block of 2048 elements with other code than its body: in the tail
// hipcc contracts x/2 * (1 + erf(x / sqrt 2)) into fma(x/2, erf, x/2) -- gelu(-6.7) is +0.0 there and -0.0 in the body, ~20 % of all
// fp16 inputs differ in the last bit (tools/micro/gelu_variants.py).  Which rows of a batch are "the tail" depends on how many
// samples share the forward: a sample's last rows get other bits alone than in a group -- the one elementwise op of the blocks that
// is not batch-invariant in torch.  Here every element takes the BODY's arithmetic (all 65 536 fp16 / bf16 inputs equal torch's body
// bit for bit: tests/test_gelu_gpu.py), whatever its position.  HBM-bound: 16 B per lane and step, 2 + 2 bytes per element.
#include "common.hpp"
#include "mfma.hpp"

namespace vlmc {

// at::native::GeluCUDAKernelImpl: opmath float;  none: x * 0.5 * (1 + erf(x * M_SQRT1_2));
// tanh: 0.5 * x * (1 + tanh(kBeta * (x + kKappa * x^3))), kBeta = M_SQRT2 * M_2_SQRTPI * 0.5, kKappa = 0.044715
template <typename T, int TANH> __device__ __forceinline__ uint16_t gelu_one(uint16_t v) {
    const float x = to_f32<T>(v);
    float y;
    if (TANH) {
        constexpr float kBeta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f, kKappa = 0.044715f;
        const float x3 = ieee_mul(ieee_mul(x, x), x);
        const float inner = ieee_mul(kBeta, ieee_add(x, ieee_mul(kKappa, x3)));
        y = ieee_mul(ieee_mul(0.5f, x), ieee_add(1.0f, tanhf(inner)));
    } else {
        y = ieee_mul(ieee_mul(x, 0.5f), ieee_add(1.0f, erff(ieee_mul(x, 0.70710678118654752440f))));
    }
    asm volatile("" : "+v"(y));                                            // (an fp32 value of its own, rounded to the dtype afterwards)
    return from_f32<T>(y);
}

template <typename T, int TANH>
__global__ __launch_bounds__(256) void gelu_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int64_t n, int vec) {
    const int64_t chunks = (n + 7) / 8;
    for (int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x; c < chunks; c += int64_t(gridDim.x) * 256) {
        const int64_t e0 = c * 8;
        uint16_t e[8];
        if (vec && e0 + 8 <= n) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t *>(x + e0);
            __builtin_memcpy(e, &v, 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = gelu_one<T, TANH>(e[j]);
            u32x4_t o;
            __builtin_memcpy(&o, e, 16);
            *reinterpret_cast<u32x4_t *>(y + e0) = o;
        } else {
            for (int j = 0; j < 8 && e0 + j < n; ++j) y[e0 + j] = gelu_one<T, TANH>(x[e0 + j]);
        }
    }
}

// fp32 tensors (the reference's Q-Former stays in fp32, blip2_t5_instruct.py:76-95): the same body arithmetic, nothing to round
template <int TANH>
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n, int vec) {
    const int64_t chunks = (n + 3) / 4;
    for (int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x; c < chunks; c += int64_t(gridDim.x) * 256) {
        const int64_t e0 = c * 4;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        const bool whole = vec && e0 + 4 <= n;
        if (whole) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t *>(x + e0);
            __builtin_memcpy(e, &v, 16);
        } else {
            for (int j = 0; j < 4 && e0 + j < n; ++j) e[j] = x[e0 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = e[j];
            float r;
            if (TANH) {
                constexpr float kBeta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f, kKappa = 0.044715f;
                const float x3 = ieee_mul(ieee_mul(v, v), v);
                r = ieee_mul(ieee_mul(0.5f, v), ieee_add(1.0f, tanhf(ieee_mul(kBeta, ieee_add(v, ieee_mul(kKappa, x3))))));
            } else {
                r = ieee_mul(ieee_mul(v, 0.5f), ieee_add(1.0f, erff(ieee_mul(v, 0.70710678118654752440f))));
            }
            e[j] = r;
        }
        if (whole) {
            u32x4_t o;
            __builtin_memcpy(&o, e, 16);
            *reinterpret_cast<u32x4_t *>(y + e0) = o;
        } else {
            for (int j = 0; j < 4 && e0 + j < n; ++j) y[e0 + j] = e[j];
        }
    }
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_gelu(const void *x, void *y, int64_t n, int dtype, int tanh_approx, void *stream) {
    VLMC_REQUIRE(dtype == VLMC_F16 || dtype == VLMC_BF16 || dtype == VLMC_F32, "vlmc_gelu: dtype must be VLMC_F16, VLMC_BF16 or VLMC_F32");
    VLMC_REQUIRE(x && y && n >= 0, "vlmc_gelu: null pointer or negative size");
    VLMC_REQUIRE(tanh_approx == 0 || tanh_approx == 1, "vlmc_gelu: tanh_approx must be 0 (erf) or 1 (tanh)");
    if (n == 0) return VLMC_OK;
    const int vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
    int64_t blocks = ((n + 7) / 8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 grid{unsigned(blocks)}, block{256};
    hipStream_t s = as_stream(stream);
    if (dtype == VLMC_F32) {
        if (tanh_approx) hipLaunchKernelGGL((gelu_f32_kernel<1>), grid, block, 0, s, static_cast<const float *>(x), static_cast<float *>(y), n, vec);
        else hipLaunchKernelGGL((gelu_f32_kernel<0>), grid, block, 0, s, static_cast<const float *>(x), static_cast<float *>(y), n, vec);
        VLMC_HIP_CHECK_LAUNCH("vlmc_gelu");
        return VLMC_OK;
    }
    const uint16_t *xi = static_cast<const uint16_t *>(x);
    uint16_t *yo = static_cast<uint16_t *>(y);
    if (dtype == VLMC_F16) {
        if (tanh_approx) hipLaunchKernelGGL((gelu_kernel<f16_t, 1>), grid, block, 0, s, xi, yo, n, vec);
        else hipLaunchKernelGGL((gelu_kernel<f16_t, 0>), grid, block, 0, s, xi, yo, n, vec);
    } else {
        if (tanh_approx) hipLaunchKernelGGL((gelu_kernel<bf16_t, 1>), grid, block, 0, s, xi, yo, n, vec);
        else hipLaunchKernelGGL((gelu_kernel<bf16_t, 0>), grid, block, 0, s, xi, yo, n, vec);
    }
    VLMC_HIP_CHECK_LAUNCH("vlmc_gelu");
    return VLMC_OK;
}
