// Micro-benchmark: the floor of the REGISTER-RESIDENT matrix-wide select (matrix_fused_kernel).
// One 1024-lane workgroup per CU; every lane loads its R 16-byte chunks of W once, the workgroups meet at B grid barriers
// (arrival counter + bounded spin, device-scope atomics -- the barrier of the product kernel), then every lane stores its
// chunks (half the weights zeroed) and 8 mask bytes per chunk.  No sampling, no histogram, no selection: what is left is
// load -> exchange(s) -> store, which no kernel of this design can beat.  FLUSH=1 adds the 2048-bin histogram flush
// (two bins per 64-bit atomic) before the first barrier.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/resident_floor.hip -o tools/micro/resident_floor && tools/micro/resident_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int R = 14;

__device__ __forceinline__ uint32_t ld_dev(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void grid_barrier(uint32_t *ctr, uint32_t n) {
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v = atomicAdd(ctr, 1u) + 1u;
        for (uint32_t it = 0; v < n && it < (1u << 14); ++it) {   // bounded: every wave reaches the end whatever the residency
            __builtin_amdgcn_s_sleep(8);
            v = ld_dev(ctr);
        }
    }
    __syncthreads();
}

__global__ void fill(uint32_t *p, size_t n) {                   // spread bit patterns: the LDS histogram must not see one bin only
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        uint32_t h = uint32_t(i) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = h;
    }
}

template <int B, int FLUSH, int COPIES = 1, int STAGGER = 0, int NATOM = 1024>
__global__ __launch_bounds__(1024, 1) void resident(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u32x2 *__restrict__ mask,
                                                    size_t chunks, uint32_t *ctl, unsigned long long *hist) {
    __shared__ uint32_t lh[2048];
    const size_t per_wg = (chunks + gridDim.x - 1) / gridDim.x;
    const size_t c0 = size_t(blockIdx.x) * per_wg, c1 = c0 + per_wg < chunks ? c0 + per_wg : chunks;
    for (int i = threadIdx.x; i < 2048; i += 1024) lh[i] = 0;
    __syncthreads();
    u32x4 v[R];
    uint32_t acc = 0;
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const size_t c = c0 + size_t(s) * 1024 + threadIdx.x;
        v[s] = c < c1 ? __builtin_nontemporal_load(in + c) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int s = 0; s < R; ++s) {
        acc += v[s].x ^ v[s].w;
        if (FLUSH) atomicAdd(&lh[(v[s].y >> 3) & 2047u], 1u);
    }
    if (FLUSH) {
        __syncthreads();
        const int i = STAGGER ? (threadIdx.x + blockIdx.x * 61) & 1023 : threadIdx.x;      // two bins per 64-bit atomic
        const unsigned long long two = (unsigned long long)lh[2 * i] | ((unsigned long long)lh[2 * i + 1] << 32);
        if (two && threadIdx.x < NATOM) atomicAdd(hist + (blockIdx.x % COPIES) * 1024 + i, two);
    }
    if (B >= 1) grid_barrier(ctl + 0, gridDim.x);
    uint32_t thr = acc;
    if (B >= 1 && FLUSH)
        for (int c = 0; c < COPIES; ++c) thr += uint32_t(__hip_atomic_load(hist + c * 1024 + (threadIdx.x & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (B >= 2) {
        if (threadIdx.x < 64) __hip_atomic_store(ctl + 64 + blockIdx.x * 64 + threadIdx.x, thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier(ctl + 1, gridDim.x);
        thr += ld_dev(ctl + 64 + ((blockIdx.x + 1) % gridDim.x) * 64 + (threadIdx.x & 63));
    }
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const size_t c = c0 + size_t(s) * 1024 + threadIdx.x;
        if (c >= c1) continue;
        u32x4 w = v[s];
        u32x2 m;
        m.x = (w.x & 0x01010101u) | (thr & 1u);
        m.y = (w.y & 0x01010101u);
        w.x &= 0xFFFF0000u; w.z &= 0x0000FFFFu;
        __builtin_nontemporal_store(m, mask + c);
        __builtin_nontemporal_store(w, out + c);
    }
}

template <int B, int FLUSH, int COPIES = 1, int STAGGER = 0, int NATOM = 1024>
static void run(const char *name, u32x4 **in, u32x4 **out, u32x2 **mask, size_t chunks, uint32_t *ctl, unsigned long long *hist, int sets) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int r = 0; r < 28; ++r) {
        const int i = r % sets;
        hipMemsetAsync(ctl, 0, 4 * (64 + 256 * 64));
        hipMemsetAsync(hist, 0, 8 * 1024 * 32);
        hipEventRecord(a);
        resident<B, FLUSH, COPIES, STAGGER, NATOM><<<256, 1024>>>(in[i], out[i], mask[i], chunks, ctl, hist);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (r >= 4) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double us = t[t.size() / 2] * 1e3, bytes = double(chunks) * 40;
    printf("%-34s median %6.1f us  min %6.1f us   %5.2f TB/s (5 B / weight)\n", name, us, t[0] * 1e3, bytes / us * 1e-6);
}

#include <algorithm>
int main() {
    const size_t elems = 25227264;                               // weights of a ViT-g block: 1408 x (4224 + 1408 + 6144 + 6144)
    const size_t chunks = elems / 8;
    const int sets = 4;                                          // rotate: 4 x 126 MB > Infinity Cache
    u32x4 *in[sets], *out[sets];
    u32x2 *mask[sets];
    for (int i = 0; i < sets; ++i) {
        hipMalloc(&in[i], chunks * 16); hipMalloc(&out[i], chunks * 16); hipMalloc(&mask[i], chunks * 8);
        fill<<<1024, 256>>>(reinterpret_cast<uint32_t *>(in[i]), chunks * 4);
    }
    uint32_t *ctl;
    unsigned long long *hist;
    hipMalloc(&ctl, 4 * (64 + 256 * 64));
    hipMalloc(&hist, 8 * 1024 * 32);
    printf("chunks per lane: %.2f of %d\n", double(chunks) / (256.0 * 1024.0), R);
    run<0, 0>("load -> store", in, out, mask, chunks, ctl, hist, sets);
    run<1, 0>("load -> barrier -> store", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1>("load -> flush + barrier -> store", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 1, 1>("  same, staggered bins", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 4, 0>("  same, 4 histogram copies", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 8, 0>("  same, 8 histogram copies", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 8, 1>("  same, 8 copies, staggered", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 32, 1>("  same, 32 copies, staggered", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 1, 0, 512>("  same, 512 atomics / workgroup", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 1, 0, 128>("  same, 128 atomics / workgroup", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 1, 0, 16>("  same, 16 atomics / workgroup", in, out, mask, chunks, ctl, hist, sets);
    run<1, 1, 1, 0, 0>("  same, LDS histogram only", in, out, mask, chunks, ctl, hist, sets);
    run<2, 0>("load -> 2 barriers -> store", in, out, mask, chunks, ctl, hist, sets);
    run<2, 1>("load -> flush + 2 barriers -> store", in, out, mask, chunks, ctl, hist, sets);
    return 0;
}
