// fp32 products of a replayed forward on fp32 matrix cores (v_mfma_f32_32x32x2_f32), BATCH-INVARIANT like their 16-bit siblings.
//
// The reference keeps the Q-Former (and `ln_vision`, `t5_proj`) in fp32 and calls them outside autocast (blip2_t5_instruct.py:76-95,
// :143-175): its linears and the two batched products of its attention (Qformer.py:201-246) are fp32 GEMMs.  A GEMM library picks its
// kernel by problem size, so a calibration sample forwarded alone and inside a stacked batch would get other last bits -- and the
// stacked capture of vlm-compression_amd/lavis/compression/pruners/calibration.py is only taken when it reproduces a sample's own forward
// bit for bit.  Here an output element is ONE fp32 accumulator that takes k in ascending pairs through one MFMA shape, whatever else
// shares the launch:
//   * vlmc_linear_fwd (dtype VLMC_F32)   Y[m][n] = sum_k X[m][k] W[n][k] + bias[n]          ("NT": both operands k-contiguous)
//   * vlmc_attn_matmul (dtype VLMC_F32)  C[b][h] = A[b][h] @ B[b][h]  through element strides (q @ k^T: B's rows are keys, k-contiguous;
//                                         probs @ v: B's rows are k, n-contiguous)
// One skeleton: a workgroup owns a 128 x 128 tile of the output, each of its four waves 64 x 64 of it (2 x 2 MFMA tiles); K goes through
// LDS in chunks of 32, double buffered (the next chunk's global loads are in flight during this chunk's 64 MFMAs per wave); both
// operand chunks lie in LDS as [row][k] with an odd pitch (the MFMA operand of a lane is one float: row = lane % 32, k = lane / 32).
// fp32 matrix peak is 157 TFLOP/s (1 / 16 of the 16-bit rate): these products are a few TFLOP per prune, not its hot path.
#include "common.hpp"

namespace vlmc {
namespace {

typedef float f32x4v_t __attribute__((ext_vector_type(4)));
typedef float f32x16v_t __attribute__((ext_vector_type(16)));
constexpr int kFT = 128, kFK = 32, kFLd = kFK + 1, kFBuf = 2 * kFT * kFLd;            // floats per LDS buffer: [A chunk | B chunk]

struct F32Gemm {
    const float *A, *B;          // A [M, K]: element (m, k) at A[m * lda + k * ska]; B: element (n, k) at B[n * sbn + k * sbk]
    float *C;                    // C [M, N]: element (m, n) at C[m * ldc + n]
    const float *bias;           // [N] or NULL
    int64_t lda, ska, sbn, sbk, ldc;
    int64_t batchA, batchB, batchC;                  // element strides between the matrices of a batch (grid z)
    int64_t b1, batchA1, batchB1, batchC1;           // z = z0 * b1 + z1: second batch level (heads): strides of z1; b1 = 1: none
    int M, N, K;
};

__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const F32Gemm g) {
    extern __shared__ float fsh[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * kFT, n0 = blockIdx.x * kFT;
    const int64_t z0 = blockIdx.z / g.b1, z1 = blockIdx.z - z0 * g.b1;
    const float *A = g.A + z0 * g.batchA + z1 * g.batchA1, *B = g.B + z0 * g.batchB + z1 * g.batchB1;
    float *C = g.C + z0 * g.batchC + z1 * g.batchC1;
    // staging: thread t takes, of each operand chunk, rows (t >> 3) + 32 i (i = 0..3), k = 4 (t & 7) .. + 3 when the operand is
    // k-contiguous (16-byte loads), or -- B of probs @ v: n-contiguous -- k = (t >> 5) + 8 i, rows 4 (t & 31) .. + 3
    const bool a_vec = g.ska == 1 && (g.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15u) == 0;
    const bool b_kcontig = g.sbk == 1;
    const bool b_vec = (b_kcontig ? (g.sbn & 3) == 0 : (g.sbk & 3) == 0 && g.sbn == 1) && (reinterpret_cast<uintptr_t>(B) & 15u) == 0;
    f32x4v_t sa[4], sb[4];
    auto fetch = [&](int q) {
        const int k0 = q * kFK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + (tid >> 3) + 32 * i, k = k0 + 4 * (tid & 7);
            f32x4v_t v = {0.f, 0.f, 0.f, 0.f};
            if (row < g.M && k < g.K) {
                const float *src = A + int64_t(row) * g.lda + int64_t(k) * g.ska;
                if (a_vec && k + 3 < g.K) v = *reinterpret_cast<const f32x4v_t *>(src);
                else
                    for (int t = 0; t < 4 && k + t < g.K; ++t) v[t] = src[int64_t(t) * g.ska];
            }
            sa[i] = v;
        }
        if (b_kcontig) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = n0 + (tid >> 3) + 32 * i, k = k0 + 4 * (tid & 7);
                f32x4v_t v = {0.f, 0.f, 0.f, 0.f};
                if (row < g.N && k < g.K) {
                    const float *src = B + int64_t(row) * g.sbn + k;
                    if (b_vec && k + 3 < g.K) v = *reinterpret_cast<const f32x4v_t *>(src);
                    else
                        for (int t = 0; t < 4 && k + t < g.K; ++t) v[t] = src[t];
                }
                sb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + (tid >> 5) + 8 * i, row = n0 + 4 * (tid & 31);
                f32x4v_t v = {0.f, 0.f, 0.f, 0.f};
                if (k < g.K && row < g.N) {
                    const float *src = B + int64_t(k) * g.sbk + int64_t(row) * g.sbn;
                    if (b_vec && row + 3 < g.N) v = *reinterpret_cast<const f32x4v_t *>(src);
                    else
                        for (int t = 0; t < 4 && row + t < g.N; ++t) v[t] = src[int64_t(t) * g.sbn];
                }
                sb[i] = v;
            }
        }
    };
    auto stash = [&](int buf) {
        float *pa = fsh + buf * kFBuf, *pb = pa + kFT * kFLd;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = pa + ((tid >> 3) + 32 * i) * kFLd + 4 * (tid & 7);
            d[0] = sa[i][0], d[1] = sa[i][1], d[2] = sa[i][2], d[3] = sa[i][3];
        }
        if (b_kcontig) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *d = pb + ((tid >> 3) + 32 * i) * kFLd + 4 * (tid & 7);
                d[0] = sb[i][0], d[1] = sb[i][1], d[2] = sb[i][2], d[3] = sb[i][3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *d = pb + (4 * (tid & 31)) * kFLd + (tid >> 5) + 8 * i;                 // rows n, column k
                d[0] = sb[i][0], d[kFLd] = sb[i][1], d[2 * kFLd] = sb[i][2], d[3 * kFLd] = sb[i][3];
            }
        }
    };
    const int rb = wave >> 1, cb = wave & 1;
    f32x16v_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nq = (g.K + kFK - 1) / kFK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) fetch(q + 1);
        const float *pa = fsh + (q & 1) * kFBuf, *pb = pa + kFT * kFLd;
        const float *ap0 = pa + (rb * 64 + (lane & 31)) * kFLd + (lane >> 5), *ap1 = ap0 + 32 * kFLd;
        const float *bp0 = pb + (cb * 64 + (lane & 31)) * kFLd + (lane >> 5), *bp1 = bp0 + 32 * kFLd;
#pragma unroll
        for (int k = 0; k < kFK; k += 2) {
            const float a0 = ap0[k], a1 = ap1[k], b0 = bp0[k], b1 = bp1[k];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (q + 1 < nq) {
            stash((q + 1) & 1);
            __syncthreads();
        }
    }
    // register r of a tile: row 8 (r / 4) + 4 (lane / 32) + r % 4 of A's rows (m), column lane % 32 of B's rows (n)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + cb * 64 + j * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bv = g.bias != nullptr ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + rb * 64 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (m < g.M) C[int64_t(m) * g.ldc + n] = g.bias != nullptr ? ieee_add(acc[i][j][r], bv) : acc[i][j][r];
            }
        }
}

int launch_f32(const char *what, const F32Gemm &g, int64_t batches, hipStream_t s) {
    if (g.M == 0 || g.N == 0 || batches == 0) return VLMC_OK;
    const size_t lds = size_t(2) * kFBuf * sizeof(float);
    static PerDeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) {
            set_error("%s: cannot reserve %zu bytes of LDS", what, lds);
            return VLMC_EHIP;
        }
        once.mark(dev);
    }
    if (batches > 65535) {
        set_error("%s: more than 65535 matrices in a batch", what);
        return VLMC_EINVAL;
    }
    const dim3 grid{unsigned((g.N + kFT - 1) / kFT), unsigned((g.M + kFT - 1) / kFT), unsigned(batches)};
    VLMC_LAUNCH_TIMED_LDS(gemm_f32_kernel, grid, dim3(256), lds, s, g);
    VLMC_HIP_CHECK_LAUNCH(what);
    return VLMC_OK;
}

}  // namespace

// Y = X W^T + bias, fp32 (called by vlmc_linear_fwd for VLMC_F32)
int linear_fwd_f32(const void *X, const void *W, const void *bias, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw, void *Y,
                   int64_t ldy, hipStream_t s) {
    VLMC_REQUIRE(X && W && Y, "vlmc_linear_fwd: null pointer");
    VLMC_REQUIRE(M >= 0 && N > 0 && K > 0 && M < (int64_t(1) << 31) && N < (int64_t(1) << 31) && K < (int64_t(1) << 31), "vlmc_linear_fwd: bad shape");
    VLMC_REQUIRE(ldx >= K && ldw >= K && ldy >= N, "vlmc_linear_fwd: a row stride is shorter than its row");
    F32Gemm g{};
    g.A = static_cast<const float *>(X), g.B = static_cast<const float *>(W), g.C = static_cast<float *>(Y);
    g.bias = static_cast<const float *>(bias);
    g.lda = ldx, g.ska = 1, g.sbn = ldw, g.sbk = 1, g.ldc = ldy;
    g.b1 = 1;
    g.M = int(M), g.N = int(N), g.K = int(K);
    return launch_f32("vlmc_linear_fwd", g, 1, s);
}

// C[b0][b1] = A[b0][b1] @ B[b0][b1], fp32, operands through element strides (called by vlmc_attn_matmul for VLMC_F32)
int attn_matmul_f32(const void *A, const void *B, void *C, int64_t batch0, int64_t batch1, int64_t M, int64_t N, int64_t K, int64_t sa_b0,
                    int64_t sa_b1, int64_t sa_m, int64_t sa_k, int64_t sb_b0, int64_t sb_b1, int64_t sb_k, int64_t sb_n, int64_t sc_b0,
                    int64_t sc_b1, int64_t sc_m, hipStream_t s) {
    VLMC_REQUIRE(A && B && C, "vlmc_attn_matmul: null pointer");
    VLMC_REQUIRE(batch0 > 0 && batch1 > 0 && M > 0 && N > 0 && K > 0 && M < (int64_t(1) << 31) && N < (int64_t(1) << 31) && K < (int64_t(1) << 31),
                 "vlmc_attn_matmul: bad shape");
    VLMC_REQUIRE(sb_k == 1 || sb_n == 1, "vlmc_attn_matmul (fp32): B must be contiguous along k or along n");
    F32Gemm g{};
    g.A = static_cast<const float *>(A), g.B = static_cast<const float *>(B), g.C = static_cast<float *>(C);
    VLMC_REQUIRE(sc_m >= N, "vlmc_attn_matmul: C's row stride is shorter than its row");
    g.lda = sa_m, g.ska = sa_k, g.sbn = sb_n, g.sbk = sb_k, g.ldc = sc_m;
    g.batchA = sa_b0, g.batchB = sb_b0, g.batchC = sc_b0;
    g.b1 = batch1, g.batchA1 = sa_b1, g.batchB1 = sb_b1, g.batchC1 = sc_b1;
    g.M = int(M), g.N = int(N), g.K = int(K);
    return launch_f32("vlmc_attn_matmul", g, batch0 * batch1, s);
}

}  // namespace vlmc
