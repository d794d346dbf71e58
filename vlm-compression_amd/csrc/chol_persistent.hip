// vlmc_chol_inverse: Cholesky factor AND its inverse of a symmetric positive definite fp32 matrix in ONE persistent launch.
//
// SparseGPT needs the upper Cholesky factor of H^-1 (sparsegpt_pruner.py:112-150: cholesky -> cholesky_inverse -> cholesky).
// vlmc/sparsegpt.py obtains it from ONE factorization of the index-reversed Hessian: J H J = M M^T, U = J M^-1 J.  Until
// round 3 that was a chain of launches per 128 columns -- a one-workgroup diagonal-block kernel (77 us), the panel below it,
// the block row of the inverse and the trailing update as library GEMMs, ~190 us per step with the GPU idle most of the time:
// 9.1 ms for n = 6144, the longest wait of every transformer block.  Here the whole computation is ONE grid of persistent
// workgroups that draw TILE TASKS (128 x 128 tiles, left-looking) from a ticket counter:
//
//   D(j)     S = A[j][j] - sum_{m<j} M[j][m] M[j][m]^T;  M[j][j] = chol(S), X[j][j] = M[j][j]^-1       (in LDS, chol_block.hpp)
//   T(i, j)  S = A[i][j] - sum_{m<j} M[i][m] M[j][m]^T;  M[i][j] = S X[j][j]^T                           i > j
//   I(j, c)  P = sum_{m=c}^{j-1} M[j][m] X[m][c];        X[j][c] = -X[j][j] P                             c < j   (X = M^-1)
//
// in the order D(j), T(j+1.., j), I(j, 0..j-1), j = 0, 1, ..: every input of a task is the output of a task with a LOWER
// ticket, so whichever workgroups are resident make progress -- no grid barrier, no co-residency requirement; a workgroup
// waits for an input tile by polling that tile's flag (bounded; a give-up aborts every workgroup and reports through info).
// A task accumulates its sum in ascending m while its inputs become ready, so when the last one arrives one 128^3 product
// is left: the critical path per 128 columns is the diagonal block's factorization + two products instead of five dependent
// launches.  Products run on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate), operands staged in LDS.
// Hand-off between workgroups (8 XCDs with private L2s): tiles are stored write-through (sc1) and drained before ONE lane
// stores the flag with an agent-scope atomic; consumers poll with agent-scope atomic loads and read tiles with sc1 loads only
// (cdna_hip_programming.md Guideline 16).  The summation order of every tile is fixed (ascending m, one accumulator per
// element): the result does not depend on how many workgroups run or in which order they draw tickets.
#include "common.hpp"
#include "chol_block.hpp"

namespace vlmc {
namespace {

constexpr int NB = kCholNb;                 // 128
constexpr int LD = kCholLd;                 // 129 floats: column walks of 32 lanes hit 32 banks
constexpr int NTH = kCholThreads;           // 512 = 8 waves
constexpr int BUF = NB * LD;                // floats per LDS tile
constexpr unsigned SPIN_LIMIT = 1u << 22;   // polls of one flag before giving up (~ seconds; a wait is normally < 1 ms)

typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct PArgs {
    const float *A;                         // [n][lda] the matrix (lower tiles read)
    float *M, *X;                           // [n][ldm] factor, [n][ldx] its inverse (lower tiles written; X's upper tiles untouched)
    int64_t lda, ldm, ldx;
    int nblk;
    int *info;
    unsigned *ws;                           // [0] ticket counter, [1] abort word, [2 ..] flags of M's tiles, then of X's tiles
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}

// ---- tile <-> LDS -------------------------------------------------------------------------------------------------------
// a 128 x 128 fp32 tile at `base` (row stride ld floats) into `dst` [128][LD]; SC1: the tile was written by another
// workgroup of this launch (every such load bypasses the caches that are not coherent across CUs / XCDs)
template <bool SC1> __device__ __forceinline__ void load_tile(float *dst, const float *base, int64_t ld, int tid) {
    const __amdgpu_buffer_rsrc_t r = rsrc_of(base);
    u32x4_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * NTH, row = e >> 5, c4 = (e & 31) * 4;
        const int off = int((int64_t(row) * ld + c4) * 4);
        v[i] = SC1 ? __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * NTH, row = e >> 5, c4 = (e & 31) * 4;
        float *d = dst + row * LD + c4;
        d[0] = __uint_as_float(v[i][0]);
        d[1] = __uint_as_float(v[i][1]);
        d[2] = __uint_as_float(v[i][2]);
        d[3] = __uint_as_float(v[i][3]);
    }
}

// `src` [128][LD] -> the tile at `base`, write-through; lower == true: elements above the diagonal are written as zeros
__device__ __forceinline__ void store_tile_sc1(float *base, int64_t ld, const float *src, int tid, bool lower, float sign) {
    const __amdgpu_buffer_rsrc_t r = rsrc_of(base);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * NTH, row = e >> 5, c4 = (e & 31) * 4;
        const float *s = src + row * LD + c4;
        u32x4_t v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = __float_as_uint((lower && c4 + t > row) ? 0.f : sign * s[t]);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, int((int64_t(row) * ld + c4) * 4), 0, 16);
    }
}

// ---- 128 x 128 x 128 product on v_mfma_f32_32x32x2_f32 -------------------------------------------------------------------
// wave w owns rows 32 (w >> 1) .. and the two 32-column blocks 64 (w & 1) .., + 32: acc[t][r] = C[32 rb + 8 (r / 4) + 4 (lane / 32) + r % 4]
//                                                                                               [64 (w & 1) + 32 t + lane % 32]
// C += sum_k a[row][k] * (BT ? b[col][k] : b[k][col])
template <bool BT> __device__ __forceinline__ void mma_tile(f32x16_t (&acc)[2], const float *a, const float *b, int wave, int lane) {
    const int rb = wave >> 1, cb = (wave & 1) * 2;
    const float *ap = a + (rb * 32 + (lane & 31)) * LD + (lane >> 5);
    const float *bp0, *bp1;
    if (BT) {
        bp0 = b + ((cb + 0) * 32 + (lane & 31)) * LD + (lane >> 5);
        bp1 = b + ((cb + 1) * 32 + (lane & 31)) * LD + (lane >> 5);
    } else {
        bp0 = b + (lane >> 5) * LD + (cb + 0) * 32 + (lane & 31);
        bp1 = b + (lane >> 5) * LD + (cb + 1) * 32 + (lane & 31);
    }
#pragma unroll 8
    for (int k = 0; k < NB; k += 2) {
        const float av = ap[k];
        const float b0 = BT ? bp0[k] : bp0[k * LD];
        const float b1 = BT ? bp1[k] : bp1[k * LD];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
    }
}

// accumulators -> `dst` [128][LD] as  dst = base_tile - acc  (base_tile already in dst)  or  dst = acc
template <bool SUBTRACT_FROM_DST> __device__ __forceinline__ void acc_to_lds(float *dst, const f32x16_t (&acc)[2], int wave, int lane) {
    const int rb = wave >> 1, cb = (wave & 1) * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), col = (cb + t) * 32 + (lane & 31);
            float *d = dst + row * LD + col;
            *d = SUBTRACT_FROM_DST ? *d - acc[t][r] : acc[t][r];
        }
}

__device__ __forceinline__ void zero_acc(f32x16_t (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// ---- flags ----------------------------------------------------------------------------------------------------------------
// wait until *flag != 0 (bounded); false: give up (the abort word is set, every workgroup leaves)
__device__ __forceinline__ bool wait_flag(unsigned *flag, unsigned *abort_word, int *info, int tid, int *sh_ok) {
    if (tid == 0) {
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            if ((++spins & 63u) == 0u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                ok = 0;
                break;
            }
            if (spins > SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(info, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // "gave up waiting": never expected
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *sh_ok = ok;
    }
    __syncthreads();
    const bool ok = *sh_ok != 0;
    __syncthreads();                                                    // (sh_ok may be rewritten by the next wait)
    return ok;
}

// every wave has drained its write-through stores; then ONE lane raises the flag
__device__ __forceinline__ void publish(unsigned *flag, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

__global__ __launch_bounds__(NTH) void chol_inverse_persistent_kernel(const PArgs p) {
    extern __shared__ float sh[];
    float *ta = sh, *tb = sh + BUF;                                     // two operand tiles [128][LD]
    __shared__ int sh_ticket, sh_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = p.nblk;
    unsigned *ticket = p.ws, *abort_word = p.ws + 1, *fM = p.ws + 2, *fX = p.ws + 2 + nblk * nblk;
    const int total = nblk * nblk;                                      // nblk tasks per column
    auto Mt = [&](int i, int j) { return p.M + (int64_t(i) * p.ldm + j) * NB; };
    auto Xt = [&](int i, int j) { return p.X + (int64_t(i) * p.ldx + j) * NB; };
    auto At = [&](int i, int j) { return p.A + (int64_t(i) * p.lda + j) * NB; };

    for (;;) {
        if (tid == 0)                                                   // (one lane decides for the workgroup: uniform)
            sh_ticket = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u
                            ? total
                            : int(__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        const int t = sh_ticket;
        __syncthreads();
        if (t >= total) return;
        const int j = t / nblk, r = t - j * nblk;
        f32x16_t acc[2];
        zero_acc(acc);
        if (r <= nblk - 1 - j) {
            // ---- D(j) (r == 0) or T(i, j): S = A[i][j] - sum_{m < j} M[i][m] M[j][m]^T ------------------------------------
            const int i = j + r;
            for (int m = 0; m < j; ++m) {
                if (!wait_flag(fM + i * nblk + m, abort_word, p.info, tid, &sh_ok)) return;
                if (i != j && !wait_flag(fM + j * nblk + m, abort_word, p.info, tid, &sh_ok)) return;
                load_tile<true>(ta, Mt(i, m), p.ldm, tid);
                if (i != j) load_tile<true>(tb, Mt(j, m), p.ldm, tid);
                __syncthreads();
                mma_tile<true>(acc, ta, i != j ? tb : ta, wave, lane);
                __syncthreads();
            }
            load_tile<false>(ta, At(i, j), p.lda, tid);                  // the matrix itself: written before the launch
            __syncthreads();
            acc_to_lds<true>(ta, acc, wave, lane);                      // ta = S
            __syncthreads();
            if (r == 0) {
                for (int e = tid; e < NB * NB; e += NTH) {              // lower part only; v = 0
                    const int row = e >> 7, col = e & 127;
                    if (col > row) ta[row * LD + col] = 0.f;
                    tb[row * LD + col] = 0.f;
                }
                __syncthreads();
                chol_block_lds(ta, tb, NB, p.info, j * NB, tid);        // ta = M[j][j], tb = its inverse (+ scratch above)
                // a non-positive pivot: the matrix is not positive definite, nothing computed from here on is of use -- every
                // workgroup leaves (the caller retries with damping: sparsegpt_pruner.py:112-128)
                if (tid == 0 && __hip_atomic_load(p.info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                    __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                store_tile_sc1(Mt(j, j), p.ldm, ta, tid, true, 1.f);
                store_tile_sc1(Xt(j, j), p.ldx, tb, tid, true, 1.f);
                publish(fM + j * nblk + j, tid);
                publish(fX + j * nblk + j, tid);
            } else {
                if (!wait_flag(fX + j * nblk + j, abort_word, p.info, tid, &sh_ok)) return;
                load_tile<true>(tb, Xt(j, j), p.ldx, tid);
                __syncthreads();
                zero_acc(acc);
                mma_tile<true>(acc, ta, tb, wave, lane);                // M[i][j] = S X[j][j]^T
                __syncthreads();
                acc_to_lds<false>(ta, acc, wave, lane);
                __syncthreads();
                store_tile_sc1(Mt(i, j), p.ldm, ta, tid, false, 1.f);
                publish(fM + i * nblk + j, tid);
            }
        } else {
            // ---- I(j, c): P = sum_{m = c}^{j - 1} M[j][m] X[m][c];  X[j][c] = -X[j][j] P -----------------------------------
            const int c = r - (nblk - j);
            for (int m = c; m < j; ++m) {
                if (!wait_flag(fM + j * nblk + m, abort_word, p.info, tid, &sh_ok)) return;
                if (!wait_flag(fX + m * nblk + c, abort_word, p.info, tid, &sh_ok)) return;
                load_tile<true>(ta, Mt(j, m), p.ldm, tid);
                load_tile<true>(tb, Xt(m, c), p.ldx, tid);
                __syncthreads();
                mma_tile<false>(acc, ta, tb, wave, lane);
                __syncthreads();
            }
            acc_to_lds<false>(tb, acc, wave, lane);                     // tb = P as [k][n]
            if (!wait_flag(fX + j * nblk + j, abort_word, p.info, tid, &sh_ok)) return;
            load_tile<true>(ta, Xt(j, j), p.ldx, tid);
            __syncthreads();
            zero_acc(acc);
            mma_tile<false>(acc, ta, tb, wave, lane);
            __syncthreads();
            acc_to_lds<false>(ta, acc, wave, lane);
            __syncthreads();
            store_tile_sc1(Xt(j, c), p.ldx, ta, tid, false, -1.f);
            publish(fX + j * nblk + c, tid);
        }
        __syncthreads();                                                // the tiles are free for the next task
    }
}

}  // namespace vlmc

using namespace vlmc;

extern "C" size_t vlmc_chol_inverse_workspace(int64_t n) {
    if (n <= 0) return 0;
    const int64_t nblk = (n + NB - 1) / NB;
    return size_t(round_up(size_t(2 + 2 * nblk * nblk) * 4, 256));
}

extern "C" int vlmc_chol_inverse(const float *A, int64_t n, int64_t lda, float *M, int64_t ldm, float *X, int64_t ldx, int *info,
                                 void *workspace, size_t workspace_bytes, int max_workgroups, void *stream) {
    VLMC_REQUIRE(A && M && X && info && workspace, "vlmc_chol_inverse: null pointer");
    VLMC_REQUIRE(n > 0 && n % NB == 0 && n < (int64_t(1) << 20), "vlmc_chol_inverse: n must be a positive multiple of %d", NB);
    VLMC_REQUIRE(lda >= n && ldm >= n && ldx >= n && lda % 4 == 0 && ldm % 4 == 0 && ldx % 4 == 0,
                 "vlmc_chol_inverse: row strides must be >= n and multiples of 4");
    VLMC_REQUIRE(aligned16(A) && aligned16(M) && aligned16(X), "vlmc_chol_inverse: matrices must be 16-byte aligned");
    VLMC_REQUIRE(NB * lda * 4 < (int64_t(1) << 31) && NB * ldm * 4 < (int64_t(1) << 31) && NB * ldx * 4 < (int64_t(1) << 31),
                 "vlmc_chol_inverse: row stride too large for 32-bit byte offsets inside a tile");
    const size_t need = vlmc_chol_inverse_workspace(n);
    if (workspace_bytes < need) {
        set_error("vlmc_chol_inverse: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
        return VLMC_EWORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(workspace, 0, need, s) != hipSuccess) {            // ticket counter, abort word, every flag
        set_error("vlmc_chol_inverse: hipMemsetAsync failed");
        return VLMC_EHIP;
    }
    const size_t lds = size_t(2) * BUF * sizeof(float);
    static PerDeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(chol_inverse_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)) != hipSuccess) {
            set_error("vlmc_chol_inverse: cannot reserve %zu B of LDS", lds);
            return VLMC_EHIP;
        }
        once.mark(dev);
    }
    const int nblk = int(n / NB);
    PArgs p{A, M, X, lda, ldm, ldx, nblk, info, static_cast<unsigned *>(workspace)};
    int grid = nblk * nblk;
    if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;
    hipLaunchKernelGGL(chol_inverse_persistent_kernel, dim3(unsigned(grid)), dim3(NTH), lds, s, p);
    VLMC_HIP_CHECK_LAUNCH("vlmc_chol_inverse");
    return VLMC_OK;
}
