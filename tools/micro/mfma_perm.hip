// Does v_mfma_f32_16x16x32_{f16,bf16} give the same bits when the 32 k-slots are permuted (the same permutation on both operands)?
// Decides whether a fused attention may hand probabilities to the second product in another slot order than vlmc_attn_matmul uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool BF>
__global__ void k(const uint16_t *A, const uint16_t *B, const int *perm, float *out, float *outp, int chain) {
    // A: [tiles][16 m][32 k], B: [tiles][16 n][32 k]; one wave per tile; chain: accumulate `chain` consecutive tiles
    const int lane = threadIdx.x, r = lane & 15, c = lane >> 4;
    f32x4_t acc = {0, 0, 0, 0}, accp = {0, 0, 0, 0};
    for (int t = 0; t < chain; ++t) {
        const uint16_t *a = A + ((size_t)(blockIdx.x * chain + t) * 16 + r) * 32, *b = B + ((size_t)(blockIdx.x * chain + t) * 16 + r) * 32;
        uint16_t ea[8], eb[8], pa[8], pb[8];
        for (int j = 0; j < 8; ++j) {
            ea[j] = a[8 * c + j]; eb[j] = b[8 * c + j];
            pa[j] = a[perm[8 * c + j]]; pb[j] = b[perm[8 * c + j]];
        }
        if (BF) {
            bf16x8_t x, y, xp, yp;
            memcpy(&x, ea, 16); memcpy(&y, eb, 16); memcpy(&xp, pa, 16); memcpy(&yp, pb, 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xp, yp, accp, 0, 0, 0);
        } else {
            f16x8_t x, y, xp, yp;
            memcpy(&x, ea, 16); memcpy(&y, eb, 16); memcpy(&xp, pa, 16); memcpy(&yp, pb, 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_f32_16x16x32_f16(xp, yp, accp, 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; ++i) {
        out[(size_t)blockIdx.x * 256 + lane * 4 + i] = acc[i];
        outp[(size_t)blockIdx.x * 256 + lane * 4 + i] = accp[i];
    }
}

static uint16_t f2h(float f, bool bf) {
    if (bf) { uint32_t u; memcpy(&u, &f, 4); return (u + 0x7fff + ((u >> 16) & 1)) >> 16; }
    _Float16 h = (_Float16)f; uint16_t r; memcpy(&r, &h, 2); return r;
}

int main() {
    const int tiles = 4096, chain = 3, n = tiles * chain * 16 * 32;
    uint16_t *hA = (uint16_t *)malloc(n * 2), *hB = (uint16_t *)malloc(n * 2);
    int perms[3][32];
    for (int s = 0; s < 32; ++s) {
        const int c = s / 8, j = s % 8;
        perms[0][s] = (j < 4) ? 4 * c + j : 16 + 4 * c + (j - 4);              // the sdpa kernel's slot order
        perms[1][s] = 31 - s;                                                  // reversed
        perms[2][s] = (s * 5 + 3) % 32;                                        // a scramble
    }
    for (int bf = 0; bf < 2; ++bf)
        for (int dist = 0; dist < 2; ++dist)
            for (int p = 0; p < 3; ++p) {
                srand(1 + dist);
                for (int i = 0; i < n; ++i) {
                    float u = (rand() / (float)RAND_MAX) * 2 - 1, v = (rand() / (float)RAND_MAX) * 2 - 1;
                    if (dist) { u *= __builtin_powif(2.f, rand() % 12 - 6); v *= __builtin_powif(2.f, rand() % 12 - 6); }   // wide dynamic range
                    hA[i] = f2h(u, bf); hB[i] = f2h(dist ? v : fabsf(v) * 0.01f, bf);    // dist 0: B like probabilities
                }
                uint16_t *dA, *dB; int *dp; float *o, *op;
                hipMalloc(&dA, n * 2); hipMalloc(&dB, n * 2); hipMalloc(&dp, 128); hipMalloc(&o, tiles * 1024); hipMalloc(&op, tiles * 1024);
                hipMemcpy(dA, hA, n * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB, n * 2, hipMemcpyHostToDevice);
                hipMemcpy(dp, perms[p], 128, hipMemcpyHostToDevice);
                if (bf) hipLaunchKernelGGL(k<true>, dim3(tiles), dim3(64), 0, 0, dA, dB, dp, o, op, chain);
                else hipLaunchKernelGGL(k<false>, dim3(tiles), dim3(64), 0, 0, dA, dB, dp, o, op, chain);
                float *ho = (float *)malloc(tiles * 1024), *hop = (float *)malloc(tiles * 1024);
                hipMemcpy(ho, o, tiles * 1024, hipMemcpyDeviceToHost); hipMemcpy(hop, op, tiles * 1024, hipMemcpyDeviceToHost);
                long diff = 0; for (int i = 0; i < tiles * 256; ++i) diff += memcmp(&ho[i], &hop[i], 4) != 0;
                printf("%s dist %d perm %d: %ld of %d outputs differ\n", bf ? "bf16" : "f16", dist, p, diff, tiles * 256);
                hipFree(dA); hipFree(dB); hipFree(dp); hipFree(o); hipFree(op); free(ho); free(hop);
            }
    return 0;
}
