// Micro-benchmark: how fast does a CU take operand tiles from L2 / HBM into LDS with global_load_lds_dwordx4 when a tile row
// contributes 64 B per request (the GEMM's 32-wide K-steps) or a whole 128-byte line (64-wide K-steps)?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_dma_rows.hip -o /tmp/lds_dma_rows && /tmp/lds_dma_rows
// Every workgroup (512 lanes) streams the K extent of its own 512 rows (256 of "W", 256 of "X" as in a 256 x 256 GEMM tile;
// W rows are shared by the workgroups of a column of tiles, X rows by a row of tiles, so most requests hit L2) into a ring
// of LDS slots, waits for them with counted vmcnt, and computes nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16(const void *gptr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gptr) : "memory", "m0");
}

// SEG: bytes of a row per step (64 or 128); NSLOT ring slots of 512 rows x SEG bytes
template <int SEG, int NSLOT>
__global__ __launch_bounds__(512, 1) void stream_kernel(const uint16_t *W, const uint16_t *X, int64_t ld, int K, int np, int nq,
                                                         uint32_t *sink) {
    constexpr int ROWS = 512, SLOT = ROWS * SEG, LPR = SEG / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;                                         // rows per wave-instruction
    constexpr int PER_WAVE = ROWS / RPI / 8;                              // instructions per wave and step
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSLOT * SLOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, bp = tile % np, bq = tile / np;
    const uint32_t lds_base = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)lds));
    const uint16_t *src[PER_WAVE];
    uint32_t dst[PER_WAVE];
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
        const int g = wave * PER_WAVE + u;                                // group of RPI rows
        const int r = g * RPI + lane / LPR;                               // tile row 0..511
        const bool is_q = r >= 256;
        const int64_t grow = is_q ? int64_t(bq) * 256 + (r - 256) : int64_t(bp) * 256 + r;
        src[u] = (is_q ? X : W) + grow * ld + (lane % LPR) * 8;
        dst[u] = g * 1024;
    }
    const int nk = K * 2 / SEG;
    auto issue = [&](int step) {
        const uint32_t slot = lds_base + (step % NSLOT) * SLOT;
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u) glds16(src[u] + step * (SEG / 2), slot + dst[u]);
    };
    for (int st = 0; st < NSLOT - 1 && st < nk; ++st) issue(st);
    uint32_t acc = 0;
    for (int t = 0; t < nk; ++t) {
        if (t + NSLOT - 1 < nk) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * PER_WAVE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + NSLOT - 1 < nk) issue(t + NSLOT - 1);
        acc += reinterpret_cast<const uint32_t *>(lds + (t % NSLOT) * SLOT)[tid];     // touch the slot
    }
    if (acc == 0xDEADBEEF) sink[0] = acc;
}

int main() {
    const int M = 32896, N = 6144, K = 1408;                               // vit.fc1
    const int np = N / 256, nq = (M + 255) / 256;
    uint16_t *W, *X;
    uint32_t *sink;
    hipMalloc(&W, size_t(N) * K * 2);
    hipMalloc(&X, size_t(nq) * 256 * K * 2);
    hipMalloc(&sink, 64);
    hipMemset(W, 1, size_t(N) * K * 2);
    hipMemset(X, 1, size_t(nq) * 256 * K * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = double(np) * nq * 512.0 * K * 2;
    auto run = [&](auto kern, const char *name) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(np * nq), dim3(512), 0, 0, W, X, int64_t(K), K, np, nq, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-44s %8.1f us  %7.2f TB/s into LDS  (%.1f GB/s per CU)\n", name, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
        }
    };
    run(stream_kernel<64, 4>, "64 B per row and step, 4 slots (128 KB)");
    run(stream_kernel<128, 2>, "128 B per row and step, 2 slots (128 KB)");
    run(stream_kernel<64, 2>, "64 B per row and step, 2 slots (64 KB)");
    run(stream_kernel<64, 3>, "64 B per row and step, 3 slots (96 KB)");
    return 0;
}
