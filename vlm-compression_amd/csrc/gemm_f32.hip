// fp32 products of a replayed forward on fp32 matrix cores (v_mfma_f32_16x16x4_f32), BATCH-INVARIANT like their 16-bit siblings.
//
// The reference keeps the Q-Former (and `ln_vision`, `t5_proj`) in fp32 and calls them outside autocast (blip2_t5_instruct.py:76-95,
// :143-175): its linears and the two batched products of its attention (Qformer.py:201-246) are fp32 GEMMs.  A GEMM library picks its
// kernel by problem size, so a calibration sample forwarded alone and inside a stacked batch would get other last bits -- and the
// stacked capture of vlm-compression_amd/lavis/compression/pruners/calibration.py is only taken when it reproduces a sample's own forward
// bit for bit.  Here an output element is ONE fp32 accumulator that takes k in ascending groups of four through one MFMA shape, whatever else
// shares the launch:
//   * vlmc_linear_fwd (dtype VLMC_F32)   Y[m][n] = sum_k X[m][k] W[n][k] + bias[n]          ("NT": both operands k-contiguous)
//   * vlmc_attn_matmul (dtype VLMC_F32)  C[b][h] = A[b][h] @ B[b][h]  through element strides (q @ k^T: B's rows are keys, k-contiguous;
//                                         probs @ v: B's rows are k, n-contiguous)
// One skeleton: a workgroup owns a square tile of the output, each of its four waves a quarter of it (TW x TW tiles of v_mfma_f32_16x16x4);
// K goes through LDS in chunks of 32-128 (the next chunk's global loads are in flight, in registers, during this chunk's products); both
// operand chunks lie in LDS as [row][k] (the MFMA operand of a lane is one float: row = lane % 16, k = lane / 16).
// fp32 matrix peak is 157 TFLOP/s (1 / 16 of the 16-bit rate): these products are a few TFLOP per prune, not its hot path.
#include "common.hpp"

namespace vlmc {
namespace {

typedef float f32x4v_t __attribute__((ext_vector_type(4)));
// (K chunk and LDS pitch are per tile size: see gemm_f32_kernel)

struct F32Gemm {
    const float *A, *B;          // A [M, K]: element (m, k) at A[m * lda + k * ska]; B: element (n, k) at B[n * sbn + k * sbk]
    float *C;                    // C [M, N]: element (m, n) at C[m * ldc + n]
    const float *bias;           // [N] or NULL
    int64_t lda, ska, sbn, sbk, ldc;
    int64_t batchA, batchB, batchC;                  // element strides between the matrices of a batch (grid z)
    int64_t b1, batchA1, batchB1, batchC1;           // z = z0 * b1 + z1: second batch level (heads): strides of z1; b1 = 1: none
    int M, N, K;
    // a ROW MAP (vlmc_linear_fwd_rows / vlmc_linear_fwd_gather, one matrix): row m of the product (m < M) is row xrows[m] of A and row
    // yrows[m] of C; rows yrows[M .. M + n_zero) of C are written as zeros by gridDim.y's extra blocks.  NULL: row m is row m.
    const int32_t *xrows, *yrows;
    int n_zero, my;              // my: blocks along y that own products (the rest clear rows)
};

// TW = MFMA tiles (16 x 16) per wave along m and along n: the workgroup's tile is 32 TW x 32 TW (four waves, 2 x 2).  An output element
// is one accumulator of v_mfma_f32_16x16x4_f32 over k in ascending groups of four -- a serial chain of K / 4 steps of 32 cycles.  With
// few tiles what a launch costs is ONE wave's chain, so the smallest tile that still gives every SIMD a chain is taken: TW = 4
// (128 x 128) once such a tile per CU exists, else TW = 2 (64 x 64), else TW = 1 (32 x 32: M = 512, N = 768, K = 3072 is 384 chains of
// 41 us on 32 x 32 x 2 tiles of 32 x 32, 1536 chains of 10 us here).  The same MFMA shape and k order per element for every TW: the
// same bits, whatever the launch's size -- which is what makes a calibration sample's rows independent of its company.
template <int TW>
__global__ __launch_bounds__(256, (TW == 4 ? 3 : 4)) void gemm_f32_kernel(const F32Gemm g) {
    // K chunk: 128 / TW -- every thread stages four 16-byte pieces of each operand per chunk whatever the tile, so a small tile takes a long
    // chunk: what a 32 x 32 tile costs is not its products (0.1 us per 32 k) but the round trip of every chunk's loads (~1.5 us), and a
    // chunk of 128 k makes 4 x fewer of them.  LDS pitch kFK + 2: (pitch r + k) mod 32 is distinct over the 16 rows x 2 k of a half wave.
    // ONE LDS buffer (33-35 KB: four workgroups per CU -- other workgroups' products cover this one's load round trips); the next
    // chunk waits in registers while this one is multiplied, and goes to LDS between two barriers.
    constexpr int kFT = 32 * TW, kFK = 128 / TW, kFLd = kFK + 2;
    constexpr int TPR = kFK / 4, RSTEP = 256 / TPR;                               // k-contiguous operand: threads per row, rows per pass (4 passes)
    constexpr int NQ4 = kFT / 4, KSTEP = 256 / NQ4;                                // n-contiguous B: threads per k row, k rows per pass (4 passes)
    extern __shared__ float fsh[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * kFT, n0 = blockIdx.x * kFT;
    if (g.yrows != nullptr && int(blockIdx.y) >= g.my) {
        // the padding rows of Y: zeros (a padded group of ragged samples -- they stay zero through the block's row-wise ops)
        const int z0r = (int(blockIdx.y) - g.my) * kFT;
        constexpr int TPZ = kFT / 4;                                           // threads per row, four columns each
        for (int r = threadIdx.x / TPZ; r < kFT; r += 256 / TPZ) {
            const int z = z0r + r, n = n0 + 4 * (threadIdx.x % TPZ);
            if (z >= g.n_zero || n >= g.N) continue;
            float *dst = g.C + int64_t(g.yrows[g.M + z]) * g.ldc + n;
            for (int t = 0; t < 4 && n + t < g.N; ++t) dst[t] = 0.f;
        }
        return;
    }
    const int64_t z0 = blockIdx.z / g.b1, z1 = blockIdx.z - z0 * g.b1;
    const float *A = g.A + z0 * g.batchA + z1 * g.batchA1, *B = g.B + z0 * g.batchB + z1 * g.batchB1;
    float *C = g.C + z0 * g.batchC + z1 * g.batchC1;
    const bool a_vec = g.ska == 1 && (g.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15u) == 0;
    const bool b_kcontig = g.sbk == 1;
    const bool b_vec = (b_kcontig ? (g.sbn & 3) == 0 : (g.sbk & 3) == 0 && g.sbn == 1) && (reinterpret_cast<uintptr_t>(B) & 15u) == 0;
    f32x4v_t sa[4], sb[4];
    int64_t arow[4];                                                           // this thread's four rows of A (through the row map)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + (tid / TPR) + RSTEP * i;
        arow[i] = row < g.M ? (g.xrows != nullptr ? int64_t(g.xrows[row]) : int64_t(row)) : -1;
    }
    auto fetch = [&](int q) {
        const int k0 = q * kFK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 4 * (tid % TPR);
            f32x4v_t v = {0.f, 0.f, 0.f, 0.f};
            if (arow[i] >= 0 && k < g.K) {
                const float *src = A + arow[i] * g.lda + int64_t(k) * g.ska;
                if (a_vec && k + 3 < g.K) v = *reinterpret_cast<const f32x4v_t *>(src);
                else
                    for (int t = 0; t < 4 && k + t < g.K; ++t) v[t] = src[int64_t(t) * g.ska];
            }
            sa[i] = v;
        }
        if (b_kcontig) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = n0 + (tid / TPR) + RSTEP * i, k = k0 + 4 * (tid % TPR);
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
                const int k = k0 + (tid / NQ4) + KSTEP * i, row = n0 + 4 * (tid % NQ4);
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
    auto stash = [&]() {
        float *pa = fsh, *pb = pa + kFT * kFLd;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = pa + ((tid / TPR) + RSTEP * i) * kFLd + 4 * (tid % TPR);
            d[0] = sa[i][0], d[1] = sa[i][1], d[2] = sa[i][2], d[3] = sa[i][3];
        }
        if (b_kcontig) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *d = pb + ((tid / TPR) + RSTEP * i) * kFLd + 4 * (tid % TPR);
                d[0] = sb[i][0], d[1] = sb[i][1], d[2] = sb[i][2], d[3] = sb[i][3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *d = pb + (4 * (tid % NQ4)) * kFLd + (tid / NQ4) + KSTEP * i;             // rows n, column k
                d[0] = sb[i][0], d[kFLd] = sb[i][1], d[2 * kFLd] = sb[i][2], d[3 * kFLd] = sb[i][3];
            }
        }
    };
    const int rb = wave >> 1, cb = wave & 1;
    f32x4v_t acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[i][j] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
    const int nq = (g.K + kFK - 1) / kFK;
    fetch(0);
    stash();
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) fetch(q + 1);
        const float *pa = fsh, *pb = pa + kFT * kFLd;
        // operand of a lane: row lane % 16, k = lane / 16 (of the step's four)
        const float *ap = pa + (rb * 16 * TW + (lane & 15)) * kFLd + (lane >> 4);
        const float *bp = pb + (cb * 16 * TW + (lane & 15)) * kFLd + (lane >> 4);
#pragma unroll 8
        for (int k = 0; k < kFK; k += 4) {
            float a[TW], b[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) a[i] = ap[i * 16 * kFLd + k], b[i] = bp[i * 16 * kFLd + k];
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (q + 1 < nq) {
            __syncthreads();                                                  // every wave has read chunk q
            stash();
            __syncthreads();
        }
    }
    // register r of a tile: row 4 (lane / 16) + r of A's rows (m), column lane % 16 of B's rows (n)
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int n = n0 + (cb * TW + j) * 16 + (lane & 15);
            if (n >= g.N) continue;
            const float bv = g.bias != nullptr ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (rb * TW + i) * 16 + 4 * (lane >> 4) + r;
                if (m < g.M) {
                    const int64_t mp = g.yrows != nullptr ? int64_t(g.yrows[m]) : int64_t(m);
                    C[mp * g.ldc + n] = g.bias != nullptr ? ieee_add(acc[i][j][r], bv) : acc[i][j][r];
                }
            }
        }
}

template <int TW> int launch_f32_tw(const char *what, const F32Gemm &g, int64_t batches, hipStream_t s) {
    constexpr int kFT = 32 * TW, kFLd = 128 / TW + 2;
    const size_t lds = size_t(2 * kFT * kFLd) * sizeof(float);          // 33-35 KB
    F32Gemm a = g;
    a.my = (g.M + kFT - 1) / kFT;
    const int zy = g.yrows != nullptr ? (g.n_zero + kFT - 1) / kFT : 0;
    const dim3 grid{unsigned((g.N + kFT - 1) / kFT), unsigned(a.my + zy), unsigned(batches)};
    VLMC_LAUNCH_TIMED_LDS(gemm_f32_kernel<TW>, grid, dim3(256), lds, s, a);
    VLMC_HIP_CHECK_LAUNCH(what);
    return VLMC_OK;
}

int launch_f32(const char *what, const F32Gemm &g, int64_t batches, hipStream_t s) {
    if ((g.M == 0 && g.n_zero == 0) || g.N == 0 || batches == 0) return VLMC_OK;
    if (batches > 65535) {
        set_error("%s: more than 65535 matrices in a batch", what);
        return VLMC_EINVAL;
    }
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    auto tiles = [&](int t) { return int64_t((g.N + t - 1) / t) * ((g.M + t - 1) / t) * batches; };
    static const int forced = [] {
        const char *e = getenv("VLMC_F32_TILE");                       // A/B: 128 / 64 / 32 for every fp32 launch (same bits)
        return e ? atoi(e) : 0;
    }();
    if (forced == 128) return launch_f32_tw<4>(what, g, batches, s);
    if (forced == 64) return launch_f32_tw<2>(what, g, batches, s);
    if (forced == 32) return launch_f32_tw<1>(what, g, batches, s);
    // 128-tiles run three workgroups to a CU, 64-tiles four, and a "wave" of 128-tiles takes ~2.8 x as long as one of 64-tiles over the same
    // K (measured: 183 against 65 us at K = 768); a partly filled last wave costs nearly a whole one, so the count that matters is the
    // waves rounded UP -- [20480, 768] x 768: 1.25 waves of 128-tiles take 360 us, 3.75 of 64-tiles 254 (tools/bench_gemm_f32.py,
    // gpurun_out/r06/gemm_f32_tiles.log: the rule picks the faster tile in 19 of 21 shapes, within 4 % in the other two).
    const int64_t w128 = (tiles(128) + 3 * int64_t(cus) - 1) / (3 * int64_t(cus)), w64 = (tiles(64) + 4 * int64_t(cus) - 1) / (4 * int64_t(cus));
    if (tiles(128) >= cus && 14 * w128 < 5 * w64) return launch_f32_tw<4>(what, g, batches, s);
    if (tiles(64) >= cus) return launch_f32_tw<2>(what, g, batches, s);
    return launch_f32_tw<1>(what, g, batches, s);
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

// rows xrows[i] of X times W^T (+ bias) into rows yrows[i] of Y for i < n_real; rows yrows[n_real .. n_real + n_zero) of Y cleared
// (called by vlmc_linear_fwd_rows for VLMC_F32 with xrows == yrows, and by vlmc_linear_fwd_gather)
int linear_gather_f32(const void *X, const void *W, const void *bias, int64_t N, int64_t K, int64_t ldx, int64_t ldw, void *Y, int64_t ldy,
                      const int32_t *xrows, const int32_t *yrows, int64_t n_real, int64_t n_zero, hipStream_t s) {
    VLMC_REQUIRE(X && W && Y && yrows && (xrows || n_real == 0), "vlmc_linear_fwd_gather: null pointer");
    VLMC_REQUIRE(n_real >= 0 && n_zero >= 0 && n_real + n_zero >= 1 && n_real + n_zero < (int64_t(1) << 31) && N > 0 && K > 0 &&
                 N < (int64_t(1) << 31) && K < (int64_t(1) << 31), "vlmc_linear_fwd_gather: bad shape");
    VLMC_REQUIRE(ldx >= K && ldw >= K && ldy >= N, "vlmc_linear_fwd_gather: a row stride is shorter than its row");
    F32Gemm g{};
    g.A = static_cast<const float *>(X), g.B = static_cast<const float *>(W), g.C = static_cast<float *>(Y);
    g.bias = static_cast<const float *>(bias);
    g.lda = ldx, g.ska = 1, g.sbn = ldw, g.sbk = 1, g.ldc = ldy;
    g.b1 = 1;
    g.M = int(n_real), g.N = int(N), g.K = int(K);
    g.xrows = xrows, g.yrows = yrows, g.n_zero = int(n_zero);
    return launch_f32("vlmc_linear_fwd_gather", g, 1, s);
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

using namespace vlmc;

extern "C" int vlmc_linear_fwd_gather(const void *X, const void *W, const void *bias, int dtype, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                                      void *Y, int64_t ldy, const int32_t *x_rows, const int32_t *y_rows, int64_t n_real, int64_t n_zero,
                                      void *stream) {
    VLMC_REQUIRE(dtype == VLMC_F32, "vlmc_linear_fwd_gather: dtype must be VLMC_F32");
    return linear_gather_f32(X, W, bias, N, K, ldx, ldw, Y, ldy, x_rows, y_rows, n_real, n_zero, as_stream(stream));
}
