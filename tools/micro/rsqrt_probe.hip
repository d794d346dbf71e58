// Which way of forming 1 / sqrt(x) is torch.rsqrt's on this stack?  tools/rsqrt_probe.py compiles this, runs every flavour over
// random fp32 values and compares with torch.rsqrt bit for bit.
#include <hip/hip_runtime.h>
extern "C" __device__ float __ocml_rsqrt_f32(float);
extern "C" __device__ float __ocml_native_rsqrt_f32(float);
extern "C" __device__ float __ocml_sqrt_f32(float);
extern "C" __device__ float __ocml_native_sqrt_f32(float);
extern "C" __device__ float __ocml_native_recip_f32(float);

__global__ void probe(const float *x, float *y, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    switch (mode) {
        case 0: r = 1.0f / __builtin_sqrtf(v); break;
        case 1: r = rsqrtf(v); break;
        case 2: r = __builtin_amdgcn_rsqf(v); break;
        case 3: r = __frsqrt_rn(v); break;
        case 4: r = __ocml_rsqrt_f32(v); break;
        case 5: r = __ocml_native_rsqrt_f32(v); break;
        case 6: r = 1.0f / __ocml_native_sqrt_f32(v); break;
        case 7: { float y0 = __builtin_amdgcn_rsqf(v); r = y0 * (1.5f - 0.5f * v * y0 * y0); break; }
        case 8: { float y0 = __builtin_amdgcn_rsqf(v); float e = __builtin_fmaf(-v * y0, y0, 1.0f); r = __builtin_fmaf(0.5f * y0, e, y0); break; }
        case 9: r = __ocml_native_recip_f32(__builtin_sqrtf(v)); break;
        case 10: r = __builtin_amdgcn_rcpf(__builtin_sqrtf(v)); break;
        case 11: r = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v)); break;
        case 12: r = float(rsqrt(double(v))); break;
        case 13: r = float(1.0 / sqrt(double(v))); break;
        case 14: r = float(rsqrt(v)); break;                 // ::rsqrt(float): whatever overload resolution picks
        default: r = 0.f;
    }
    y[i] = r;
}
extern "C" int run_probe(const float *x, float *y, int n, int mode, void *stream) {
    hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, n, mode);
    return (int)hipGetLastError();
}
