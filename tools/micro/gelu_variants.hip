// Which arithmetic reproduces at::native::GeluCUDAKernelImpl (approximate = none) of this torch build bit for bit on 16-bit inputs?
// extern "C" gelu_variant(kind, dtype, in, out, n): out = wd(gelu(float(in)))   dtype 1 fp16, 2 bf16
#include <hip/hip_runtime.h>
#include <cstdint>
__device__ float h2f(uint16_t v) { _Float16 h; __builtin_memcpy(&h, &v, 2); return float(h); }
__device__ uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t r; __builtin_memcpy(&r, &h, 2); return r; }
__device__ float b2f(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }
__device__ uint16_t f2b(float f) { __bf16 h = (__bf16)f; uint16_t r; __builtin_memcpy(&r, &h, 2); return r; }

__device__ float g_plain(float x) {
#pragma clang fp contract(off)
    return (x * 0.5f) * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ float g_fast(float x) {
#pragma clang fp contract(fast)
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ float g_fma(float x) {                  // (x/2) + (x/2) * erf
    const float h = x * 0.5f;
    return __builtin_fmaf(h, erff(x * 0.70710678118654752440f), h);
}
__device__ float g_dbl(float x) {                  // erf in double
    return float(double(x) * 0.5 * (1.0 + erf(double(x) * 0.70710678118654752440)));
}
__device__ float g_ncdf(float x) {                 // x * normcdf(x)
    return x * normcdff(x);
}
__global__ void k(int kind, int dtype, const uint16_t *in, uint16_t *out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = dtype == 1 ? h2f(in[i]) : b2f(in[i]);
    float y = kind == 0 ? g_plain(x) : kind == 1 ? g_fast(x) : kind == 2 ? g_fma(x) : kind == 3 ? g_dbl(x) : g_ncdf(x);
    out[i] = dtype == 1 ? f2h(y) : f2b(y);
}
extern "C" void gelu_variant(int kind, int dtype, const void *in, void *out, int n) {
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, kind, dtype, (const uint16_t *)in, (uint16_t *)out, n);
    hipDeviceSynchronize();
}
