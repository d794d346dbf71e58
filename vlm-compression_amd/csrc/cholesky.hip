// K9: diagonal-block kernel of the blocked Cholesky factorization used for SparseGPT's Hessian chain
// (replaces the unblocked panel step inside torch.linalg.cholesky, sparsegpt_pruner.py:116,148).
//
// rocSOLVER's potrf spends 0.3 ms per 128-column panel in a one-workgroup unblocked kernel (6 ms for a 2048^2
// Hessian, 20 ms for 6144^2, twice per linear).  Here the 128x128 diagonal block is factorized in LDS by one
// workgroup and its inverse is produced in the same launch, so that the panel below it becomes a GEMM
// (L21 = A21 @ inv(L11)^T) and the trailing update another one -- both at the library's 100+ TFLOP/s fp32 rate.
// Right-looking column sweep; every step is elementwise IEEE fp32 (sqrt, divide, multiply, subtract).
#include "common.hpp"

#ifdef VLMC_CHOL_STAMPS                 // diagnostic build only: phase clocks (100 MHz) printed by lane 0
#define CSTAMP(i) do { if (tid == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define CSTAMP(i) do {} while (0)
#endif
#include "chol_block.hpp"

namespace vlmc {

__global__ __launch_bounds__(kCholThreads) void chol_block_kernel(const float *__restrict__ A, int64_t lda, int nb, float *__restrict__ L,
                                                         int64_t ldl, float *__restrict__ Linv, int64_t ldi,
                                                         int *__restrict__ info, int col0) {
    extern __shared__ float sh[];
    float *a = sh;                          // [128][kCholLd] block being factorized (lower part)
    float *v = sh + kCholNb * kCholLd;      // [128][kCholLd] its inverse (upper pieces double as scratch)
    const int tid = threadIdx.x;
#ifdef VLMC_CHOL_STAMPS
    __shared__ unsigned long long stamps[24];
#endif
    CSTAMP(0);
    const bool vec4 = nb == kCholNb && (lda % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0);
    if (vec4) {
        for (int e = tid; e < kCholNb * kCholNb / 4; e += kCholThreads) {
            const int i = e / (kCholNb / 4), k = (e % (kCholNb / 4)) * 4;
            const float4 q = *reinterpret_cast<const float4 *>(A + int64_t(i) * lda + k);
            const float r4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[i * kCholLd + k + t] = (k + t <= i) ? r4[t] : 0.f;
                v[i * kCholLd + k + t] = 0.f;
            }
        }
    } else {
        for (int e = tid; e < kCholNb * kCholNb; e += kCholThreads) {
            const int i = e / kCholNb, k = e % kCholNb;
            // rows / columns past nb are padded with the identity so that every 32-piece is a full, regular one
            a[i * kCholLd + k] = (i < nb && k <= i) ? A[int64_t(i) * lda + k] : ((i >= nb && k == i) ? 1.f : 0.f);
            v[i * kCholLd + k] = 0.f;
        }
    }
    __syncthreads();
    CSTAMP(1);
    chol_block_lds(a, v, nb, info, col0, tid);
    const bool vec4o = nb == kCholNb && (ldl % 4 == 0) && (ldi % 4 == 0) && (reinterpret_cast<uintptr_t>(L) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(Linv) % 16 == 0);
    if (vec4o) {                                       // full block: 16-byte stores, no division by a run-time nb
        for (int e = tid; e < kCholNb * kCholNb / 4; e += kCholThreads) {
            const int i = e / (kCholNb / 4), k = (e % (kCholNb / 4)) * 4;
            float4 ql, qv;
            ql.x = k <= i ? a[i * kCholLd + k] : 0.f;         qv.x = k <= i ? v[i * kCholLd + k] : 0.f;
            ql.y = k + 1 <= i ? a[i * kCholLd + k + 1] : 0.f; qv.y = k + 1 <= i ? v[i * kCholLd + k + 1] : 0.f;
            ql.z = k + 2 <= i ? a[i * kCholLd + k + 2] : 0.f; qv.z = k + 2 <= i ? v[i * kCholLd + k + 2] : 0.f;
            ql.w = k + 3 <= i ? a[i * kCholLd + k + 3] : 0.f; qv.w = k + 3 <= i ? v[i * kCholLd + k + 3] : 0.f;
            *reinterpret_cast<float4 *>(L + int64_t(i) * ldl + k) = ql;
            *reinterpret_cast<float4 *>(Linv + int64_t(i) * ldi + k) = qv;   // the upper part held scratch
        }
    } else {
        for (int e = tid; e < nb * nb; e += kCholThreads) {
            const int i = e / nb, k = e % nb;
            L[int64_t(i) * ldl + k] = k <= i ? a[i * kCholLd + k] : 0.f;
            Linv[int64_t(i) * ldi + k] = k <= i ? v[i * kCholLd + k] : 0.f;      // the upper part held scratch
        }
    }
#ifdef VLMC_CHOL_STAMPS
    __syncthreads();
    CSTAMP(18);
    if (tid == 0 && col0 == 0) {
        for (int i = 1; i <= 18; ++i) printf("%d:%llu ", i, (stamps[i] - stamps[0]));
        printf("\n");
    }
#endif
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_chol_block(const float *A, int64_t lda, int nb, float *L, int64_t ldl, float *Linv, int64_t ldi, int *info,
                               int col0, void *stream) {
    VLMC_REQUIRE(A && L && Linv && info, "vlmc_chol_block: null pointer");
    VLMC_REQUIRE(nb > 0 && nb <= kCholNb && lda >= nb && ldl >= nb && ldi >= nb, "vlmc_chol_block: bad block size %d (max %d)", nb,
                 kCholNb);
    const size_t lds = size_t(2) * kCholNb * kCholLd * sizeof(float);
    static PerDeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(chol_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)) != hipSuccess) {
            set_error("vlmc_chol_block: cannot reserve %zu B of LDS", lds);
            return VLMC_EHIP;
        }
        once.mark(dev);
    }
    hipLaunchKernelGGL(chol_block_kernel, dim3(1), dim3(kCholThreads), lds, as_stream(stream), A, lda, nb, L, ldl, Linv, ldi, info, col0);
    VLMC_HIP_CHECK_LAUNCH("vlmc_chol_block");
    return VLMC_OK;
}
