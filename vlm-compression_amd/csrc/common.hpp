// Shared device/host helpers for the gfx950 kernels.  Built with -ffp-contract=off:
// every fp32 operation below is a single correctly rounded IEEE op, which is what
// makes mask indices bit-identical to the reference's PyTorch path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/vlmc.h"

namespace vlmc {

// ---- error plumbing ---------------------------------------------------------------
void set_error(const char *fmt, ...);
#define VLMC_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::vlmc::set_error(__VA_ARGS__); \
            return VLMC_EINVAL;            \
        }                                  \
    } while (0)
#define VLMC_HIP_CHECK_LAUNCH(what)                                                  \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            ::vlmc::set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e_)); \
            return VLMC_EHIP;                                                        \
        }                                                                            \
    } while (0)

// ---- element types ----------------------------------------------------------------
struct f32_t { using raw = float; };
struct f16_t { using raw = uint16_t; };
struct bf16_t { using raw = uint16_t; };

template <typename T> __device__ __forceinline__ float to_f32(typename T::raw v);
template <> __device__ __forceinline__ float to_f32<f32_t>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }
template <> __device__ __forceinline__ float to_f32<f16_t>(uint16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return float(h);
}

// 8 consecutive elements of a row, as loaded by one lane (16 B for 16-bit types, 32 B for f32).
template <typename T> struct Chunk8;
template <> struct Chunk8<f32_t> { float v[8]; };
template <> struct Chunk8<f16_t> { uint16_t v[8]; };
template <> struct Chunk8<bf16_t> { uint16_t v[8]; };

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// NT = non-temporal (streaming) access: for data touched exactly once by the kernel and not re-read by the
// next one, so that it does not displace re-used lines in L2 / Infinity Cache.
template <typename T, bool NT = false> __device__ __forceinline__ Chunk8<T> load_chunk8(const typename T::raw *p) {
    Chunk8<T> c;
    const u32x4_t *q = reinterpret_cast<const u32x4_t *>(p);
    if constexpr (sizeof(typename T::raw) == 2) {
        const u32x4_t a = NT ? __builtin_nontemporal_load(q) : *q;
        __builtin_memcpy(c.v, &a, 16);
    } else {
        const u32x4_t a = NT ? __builtin_nontemporal_load(q) : q[0];
        const u32x4_t b = NT ? __builtin_nontemporal_load(q + 1) : q[1];
        __builtin_memcpy(c.v, &a, 16);
        __builtin_memcpy(c.v + 4, &b, 16);
    }
    return c;
}
template <typename T, bool NT = false> __device__ __forceinline__ void store_chunk8(typename T::raw *p, const Chunk8<T> &c) {
    u32x4_t *q = reinterpret_cast<u32x4_t *>(p);
    if constexpr (sizeof(typename T::raw) == 2) {
        u32x4_t a;
        __builtin_memcpy(&a, c.v, 16);
        if constexpr (NT) __builtin_nontemporal_store(a, q); else *q = a;
    } else {
        u32x4_t a, b;
        __builtin_memcpy(&a, c.v, 16);
        __builtin_memcpy(&b, c.v + 4, 16);
        if constexpr (NT) { __builtin_nontemporal_store(a, q); __builtin_nontemporal_store(b, q + 1); }
        else { q[0] = a; q[1] = b; }
    }
}

// Wanda score -> order-preserving unsigned key.  score = |w| * sqrt(s) >= +0 or NaN.
// NaN sorts last (torch.sort semantics) -> 0xFFFFFFFF; -0 (impossible, but harmless) == +0.
__device__ __forceinline__ uint32_t score_key(float sc) {
    return (sc != sc) ? 0xFFFFFFFFu : (__float_as_uint(sc) & 0x7FFFFFFFu);
}

// Correctly rounded fp32 primitives.  NOTE: HIP's __fsqrt_rn / __fdiv_rn are NOT the IEEE
// operations (they lower to the approximate native instructions unless OCML rounded ops are
// enabled); plain sqrtf and `/` are, under -fhip-fp32-correctly-rounded-divide-sqrt (hipcc's
// default, also passed explicitly by the Makefile).  With -ffp-contract=off `a*b` and `a+b`
// are never fused.  tests/test_wanda_gpu.py::test_ieee_sqrt_div_fma_on_device checks all of
// them against the host's IEEE arithmetic.
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }
__device__ __forceinline__ float ieee_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float ieee_add(float a, float b) { return a + b; }

constexpr int kWave = 64;

// wave-wide sum of a 32-bit integer (result valid in every lane)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// "done once per device" flag for per-device function attributes (the LDS carve-out of a kernel is a property of the
// function ON a device); thread-safe, one bit per device ordinal.
struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    // returns true when the caller must perform the one-time work for the current device (and then call mark())
    bool needed(int *dev_out) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        *dev_out = dev;
        return dev >= 64 || !(done.load(std::memory_order_acquire) >> dev & 1);
    }
    void mark(int dev) {
        if (dev < 64) done.fetch_or(uint64_t(1) << dev, std::memory_order_release);
    }
};

// Timing hook (vlmc_set_launch_events): the NEXT timed kernel launched from this thread carries the caller's HIP
// events in its own dispatch (hipExtLaunchKernel): start / stop are the kernel's begin / end timestamps, and no marker
// packet is put between kernels (an hipEventRecord between two kernels idles the GPU for ~5 us on MI355X).
struct LaunchEvents { hipEvent_t start, stop; };
LaunchEvents take_launch_events();                 // returns the pending pair (nullptr, nullptr if none) and clears it
#define VLMC_LAUNCH_TIMED(kernel, grid, block, stream, ...)                                                  \
    do {                                                                                                     \
        const ::vlmc::LaunchEvents ev_ = ::vlmc::take_launch_events();                                       \
        if (ev_.start || ev_.stop) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev_.start, ev_.stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                \
    } while (0)

#define VLMC_LAUNCH_TIMED_LDS(kernel, grid, block, lds, stream, ...)                                          \
    do {                                                                                                     \
        const ::vlmc::LaunchEvents ev_ = ::vlmc::take_launch_events();                                       \
        if (ev_.start || ev_.stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev_.start, ev_.stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                              \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ bool aligned16_dev(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace vlmc
