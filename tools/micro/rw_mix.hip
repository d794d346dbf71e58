// Micro-benchmark: what does a PLAIN stream with the per-row select's traffic mix reach on this GPU?
// Per 8 weights: read 16 B, write 16 B (weights) + 8 B (mask bytes) -- non-temporal, 1 KiB per wave-instruction,
// no selection at all.  The select kernels cannot be faster than this.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/rw_mix.hip -o tools/micro/rw_mix && tools/micro/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int CH>
__global__ __launch_bounds__(256) void rw_mix(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u32x2 *__restrict__ mask,
                                              size_t chunks) {
    size_t base = (size_t(blockIdx.x) * 256 + threadIdx.x / 64 * 64) * CH + threadIdx.x % 64;    // a wave owns CH x 64 chunks
    u32x4 v[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s)
        if (base + s * 64 < chunks) v[s] = __builtin_nontemporal_load(in + base + s * 64);
#pragma unroll
    for (int s = 0; s < CH; ++s) {
        if (base + s * 64 >= chunks) continue;
        u32x4 w = v[s];
        u32x2 m;
        m.x = (w.x & 0x01010101u);
        m.y = (w.y & 0x01010101u);
        w.x &= 0xFFFF0000u; w.z &= 0x0000FFFFu;                   // "zero half of the weights"
        __builtin_nontemporal_store(m, mask + base + s * 64);
        __builtin_nontemporal_store(w, out + base + s * 64);
    }
}

int main() {
    const size_t elems = 56623104;                               // weights of a T5-XL decoder block
    const size_t chunks = elems / 8;
    const int sets = 4;                                          // rotate: 4 x 283 MB > Infinity Cache
    u32x4 *in[sets], *out[sets];
    u32x2 *mask[sets];
    for (int i = 0; i < sets; ++i) {
        hipMalloc(&in[i], chunks * 16); hipMalloc(&out[i], chunks * 16); hipMalloc(&mask[i], chunks * 8);
        hipMemset(in[i], 0x3c, chunks * 16);
    }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int ch : {1, 2, 4}) {
        const unsigned grid = unsigned((chunks + 256 * ch - 1) / (256 * ch));
        float best = 1e9f, tot = 0.f;
        const int reps = 24;
        for (int r = 0; r < reps + 4; ++r) {
            const int i = r % sets;
            hipEventRecord(a);
            if (ch == 1) hipLaunchKernelGGL(rw_mix<1>, dim3(grid), dim3(256), 0, 0, in[i], out[i], mask[i], chunks);
            if (ch == 2) hipLaunchKernelGGL(rw_mix<2>, dim3(grid), dim3(256), 0, 0, in[i], out[i], mask[i], chunks);
            if (ch == 4) hipLaunchKernelGGL(rw_mix<4>, dim3(grid), dim3(256), 0, 0, in[i], out[i], mask[i], chunks);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (r >= 4) { tot += ms; best = ms < best ? ms : best; }
        }
        const double bytes = double(chunks) * 40.0;
        printf("chunks per lane %d: avg %.1f us  best %.1f us  ->  %.0f GB/s avg, %.0f GB/s best (%.1f MB: 2 B read + 3 B written per weight)\n",
               ch, tot / reps * 1e3, best * 1e3, bytes / (tot / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9, bytes / 1e6);
    }
    return 0;
}
