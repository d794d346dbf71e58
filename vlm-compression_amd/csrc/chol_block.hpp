// The 128 x 128 diagonal block of the blocked Cholesky factorization IN LDS: factor and inverse (shared by chol_block_kernel,
// one workgroup per launch, and by the persistent factorization kernel of chol_persistent.hip, whose diagonal tasks call it).
// `a` [128][kCholLd]: the block (lower part read) -> its lower Cholesky factor; `v` [128][kCholLd], zero on entry -> the
// inverse of the factor (its upper pieces are scratch).  512 threads; every step elementwise IEEE fp32.
#pragma once
#include "common.hpp"

namespace vlmc {

constexpr int kCholNb = 128;
constexpr int kCholLd = kCholNb + 1;       // LDS row stride: column walks hit different banks

constexpr int kSb = 32;                    // sub-block of the INVERSE (one wave per diagonal piece)
constexpr int kFb = 16;                    // sub-block of the factorization: factorized by ONE wave in registers (no workgroup
                                           // barriers inside).  The wave-level sweep costs ~ kFb^2 per piece: eight 16-wide pieces
                                           // are 51 k cycles where four 32-wide ones were 93 k (and the forward substitutions 28 k
                                           // instead of 41 k); every element still receives its updates in ascending column order
constexpr int kCholThreads = 512;

__device__ __forceinline__ float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// (the phase clocks of chol_block_kernel's diagnostic build stamp the kernel's own phases, not the inside of this function)
#pragma push_macro("CSTAMP")
#undef CSTAMP
#define CSTAMP(i) do {} while (0)

// 512 lanes.  Per 16-column sub-block: (a) wave 0 factorizes the 16x16 diagonal piece with its rows in registers
// (pivots and column entries travel by v_readlane), (b) every row below solves its 16 entries by forward substitution,
// (c) the rest of the block gets its rank-16 update.  24 barriers per 128x128 block instead of 3 per column.
// Then inv(L): the four 32x32 diagonal inverses by one wave each, the off-diagonal pieces block-diagonal by block-diagonal.
__device__ __forceinline__ void chol_block_lds(float *a, float *v, const int nb, int *info, const int col0, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int base = 0; base < kCholNb && base < nb; base += kFb) {
        if (wave == 0) {
            float row[kFb];                 // lane i < kFb: row base + i of the diagonal piece
            const int li = lane & (kFb - 1);
#pragma unroll
            for (int k = 0; k < kFb; ++k) row[k] = (k <= li) ? a[(base + li) * kCholLd + base + k] : 0.f;
#pragma unroll
            for (int j = 0; j < kFb; ++j) {
                const float d = rl(row[j], j);
                if (lane == 0 && base + j < nb && !(d > 0.f) && *info == 0) *info = col0 + base + j + 1;   // not positive definite
                const float r = ieee_sqrt(d);
                const float lij = li > j ? ieee_div(row[j], r) : (li == j ? r : row[j]);
                row[j] = lij;
#pragma unroll
                for (int k = j + 1; k < kFb; ++k) {
                    const float lkj = rl(lij, k);
                    row[k] = row[k] - ieee_mul(lij, lkj);          // (lanes li < k: the upper part, never stored -- no predicate)
                }
            }
            if (lane < kFb) {
#pragma unroll
                for (int k = 0; k < kFb; ++k)
                    if (k <= li) a[(base + li) * kCholLd + base + k] = row[k];
            }
        }
        __syncthreads();
        const int below = base + kFb;
        for (int i = below + tid; i < kCholNb; i += kCholThreads) {      // (b) forward substitution, one row per lane
            float x[kFb];
#pragma unroll
            for (int c = 0; c < kFb; ++c) {
                float acc = a[i * kCholLd + base + c];
#pragma unroll
                for (int k = 0; k < c; ++k) acc = acc - ieee_mul(x[k], a[(base + c) * kCholLd + base + k]);
                x[c] = ieee_div(acc, a[(base + c) * kCholLd + base + c]);
            }
#pragma unroll
            for (int c = 0; c < kFb; ++c) a[i * kCholLd + base + c] = x[c];
        }
        __syncthreads();
        const int m2 = (kCholNb - below) / 2;                    // (c) rank-16 update of what is left, 2x2 tiles per lane
        for (int e = tid; e < m2 * m2; e += kCholThreads) {
            const int i = below + 2 * (e / m2), k = below + 2 * (e % m2);
            if (k <= i) {
                float acc00 = a[i * kCholLd + k], acc01 = a[i * kCholLd + k + 1];
                float acc10 = a[(i + 1) * kCholLd + k], acc11 = a[(i + 1) * kCholLd + k + 1];
#pragma unroll 8
                for (int c = 0; c < kFb; ++c) {
                    const float li0 = a[i * kCholLd + base + c], li1 = a[(i + 1) * kCholLd + base + c];
                    const float lk0 = a[k * kCholLd + base + c], lk1 = a[(k + 1) * kCholLd + base + c];
                    acc00 = acc00 - ieee_mul(li0, lk0); acc01 = acc01 - ieee_mul(li0, lk1);
                    acc10 = acc10 - ieee_mul(li1, lk0); acc11 = acc11 - ieee_mul(li1, lk1);
                }
                a[i * kCholLd + k] = acc00;
                a[(i + 1) * kCholLd + k] = acc10; a[(i + 1) * kCholLd + k + 1] = acc11;
                if (k + 1 <= i) a[i * kCholLd + k + 1] = acc01;     // (i, i+1) lies above the diagonal of the i == k tile
            }
        }
        __syncthreads();
        CSTAMP(2 + base / kFb);
    }
    // ---- inverse of the lower-triangular block ----------------------------------------------------------------
    if (wave < kCholNb / kSb) {   // diagonal pieces: wave w inverts piece w; lane c < 32 owns column c:
        //                           x_r = (delta_rc - sum_{k=c}^{r-1} L[r][k] x_k) / L[r][r]
        const int base = wave * kSb, c = lane & 31;
        float x[kSb];
#pragma unroll
        for (int r = 0; r < kSb; ++r) {
            float acc = (r == c) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < r; ++k)          // (x[k] == 0 for k < c: no predicate needed, the same bits)
                acc = acc - ieee_mul(a[(base + r) * kCholLd + base + k], x[k]);
            x[r] = (r >= c) ? ieee_div(acc, a[(base + r) * kCholLd + base + r]) : 0.f;
        }
        if (lane < kSb) {
#pragma unroll
            for (int r = 0; r < kSb; ++r) v[(base + r) * kCholLd + base + c] = x[r];
        }
    }
    __syncthreads();
    CSTAMP(14);
    constexpr int NBLK = kCholNb / kSb;
    for (int d = 1; d < NBLK; ++d) {
        // pieces (rb, cb = rb - d): T = sum_{m=cb}^{rb-1} L[rb][m] V[m][cb], kept in the (cb, rb) mirror piece of v
        const int pairs = NBLK - d;
        constexpr int H = kSb / 2;                   // 2 x 2 outputs per lane: two LDS reads feed two products each
        for (int e = tid; e < pairs * H * H; e += kCholThreads) {
            const int pr = e / (H * H), r = 2 * ((e / H) % H), c = 2 * (e % H);
            const int rb = pr + d, cb = pr;
            float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
            for (int mb = cb; mb < rb; ++mb)
#pragma unroll 8
                for (int k = 0; k < kSb; ++k) {
                    const float l0 = a[(rb * kSb + r) * kCholLd + mb * kSb + k], l1 = a[(rb * kSb + r + 1) * kCholLd + mb * kSb + k];
                    const float v0 = v[(mb * kSb + k) * kCholLd + cb * kSb + c], v1 = v[(mb * kSb + k) * kCholLd + cb * kSb + c + 1];
                    a00 = __builtin_fmaf(l0, v0, a00); a01 = __builtin_fmaf(l0, v1, a01);
                    a10 = __builtin_fmaf(l1, v0, a10); a11 = __builtin_fmaf(l1, v1, a11);
                }
            v[(cb * kSb + r) * kCholLd + rb * kSb + c] = a00;        // scratch in the upper part
            v[(cb * kSb + r) * kCholLd + rb * kSb + c + 1] = a01;
            v[(cb * kSb + r + 1) * kCholLd + rb * kSb + c] = a10;
            v[(cb * kSb + r + 1) * kCholLd + rb * kSb + c + 1] = a11;
        }
        __syncthreads();
        for (int e = tid; e < pairs * H * H; e += kCholThreads) {            // V[rb][cb] = -V[rb][rb] T
            const int pr = e / (H * H), r = 2 * ((e / H) % H), c = 2 * (e % H);
            const int rb = pr + d, cb = pr;
            float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
#pragma unroll 8
            for (int k = 0; k <= r; ++k) {
                const float l0 = v[(rb * kSb + r) * kCholLd + rb * kSb + k], l1 = v[(rb * kSb + r + 1) * kCholLd + rb * kSb + k];
                const float t0 = v[(cb * kSb + k) * kCholLd + rb * kSb + c], t1 = v[(cb * kSb + k) * kCholLd + rb * kSb + c + 1];
                a00 = __builtin_fmaf(l0, t0, a00); a01 = __builtin_fmaf(l0, t1, a01);
                a10 = __builtin_fmaf(l1, t0, a10); a11 = __builtin_fmaf(l1, t1, a11);
            }
            {                                                                 // row r + 1 has one more term: k = r + 1
                const float l1 = v[(rb * kSb + r + 1) * kCholLd + rb * kSb + r + 1];
                a10 = __builtin_fmaf(l1, v[(cb * kSb + r + 1) * kCholLd + rb * kSb + c], a10);
                a11 = __builtin_fmaf(l1, v[(cb * kSb + r + 1) * kCholLd + rb * kSb + c + 1], a11);
            }
            // (the products above read T = the (cb, rb) mirror piece; the results go to the (rb, cb) piece: no overlap)
            v[(rb * kSb + r) * kCholLd + cb * kSb + c] = -a00;
            v[(rb * kSb + r) * kCholLd + cb * kSb + c + 1] = -a01;
            v[(rb * kSb + r + 1) * kCholLd + cb * kSb + c] = -a10;
            v[(rb * kSb + r + 1) * kCholLd + cb * kSb + c + 1] = -a11;
        }
        __syncthreads();
        CSTAMP(14 + d);
    }
}
#pragma pop_macro("CSTAMP")

}  // namespace vlmc
