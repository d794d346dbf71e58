// K14 / K15 fused: the SparseLoRA layer's three big products with the masked low-rank algebra INSIDE the MFMA GEMM (gfx950).
// Replaces, for 16-bit weights under autocast, the tensor algebra + autograd of
// /root/reference/lavis/peft/src/peft/tuners/lora.py:359-382:
//
//   forward      Y  = x  W_eff^T + b         W_eff = (W + s (B A)) . M   (sparse)   |   W . M + s (B A)   (masked)
//   backward     dX = dY W_eff
//                G  = dY^T x ;  Gm = (G [. M]) s ;  dB = Gm A^T ;  dA = B^T Gm
//
// Neither W_eff nor G ever exists in HBM:
//  * vlmc_sparse_lora_fwd / vlmc_sparse_lora_bwd_input -- the W operand of the GEMM is GENERATED tile by tile on its way
//    into LDS: a wave takes a 16 x 64 piece of W and of the mask into registers (whole 128-byte lines of both), forms the
//    piece of B A with four v_mfma_f32_16x16x16 (the rank padded to 16; the MFMA's row index is permuted so that its
//    result registers line up with the 16 consecutive weights a lane holds), applies the reference's rounding chain and the
//    mask in registers and writes the W_eff piece into the LDS image the main v_mfma_f32_16x16x32 loop reads -- K-major for
//    the forward (rows of W_eff are rows of the operand), as-it-lies for the backward (rows of W_eff are the reduction index:
//    the fragments come out of ds_read_b64_tr_b16).  The activations travel by LDS-DMA.  A W_eff piece is regenerated once
//    per 256 rows of activations: 1/16 of the main loop's MFMA work and ~50 packed VALU operations per 32 MFMAs.
//  * vlmc_sparse_lora_bwd_weight -- G = dY^T x as a 128 (in) x 256 (out) tile per workgroup (both operands transposed by a
//    pre-pass: the reduction runs over tokens); the epilogue rounds the tile as the reference's GEMM output is rounded,
//    masks and scales it in registers, and contracts it with A (straight from the accumulator layout) and with B (through a
//    transposing LDS image) on v_mfma_f32_16x16x16: per-tile partial sums of dA and dB, combined in a fixed order by
//    lora_partial_reduce_kernel (no float atomics: deterministic).
//
// Tile 128 (P) x 256 (Q) x 64 (K) per 512-thread workgroup, wave = 64 x 64, three LDS slots of 48 KiB: the DMA of step
// d + 2 and the register loads of the generator's piece for step d + 3 are in flight while step d is multiplied and the
// piece for step d + 1 is generated.  Every VMEM instruction of the loop is inline asm and counted by hand (hipcc sees
// none of them: one s_waitcnt vmcnt(N) per step).
//
// Rounding (wd = the 16-bit weight dtype = the autocast dtype; the entry points refuse anything else and the caller
// takes the unfused kernels of sparse_lora.hip):
//   d1 = wd(B16 A16) ; d2 = wd(d1 s) ; sparse: wd(W + d2) M ; masked: wd(W M + d2)
//   G16 = wd(dY^T x) ; Gm = wd((G16 [M]) s) ; dA, dB: fp32 sums of exact products, rounded to wd once at the end.
#include "common.hpp"
#include "mfma.hpp"

#include <cstdlib>
#include <cstring>

namespace vlmc {
namespace {

typedef _Float16 h16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

constexpr int BP = 128, BK = 64, ROWB = 128;                  // rows of 64 elements = 128 B in every LDS image
constexpr int NTH = 512, TP = 4;                              // 2 (p) x 4 (q) waves of 64 x (BQ / 4); BQ = 256 or 192 rows of Q per tile
constexpr int P_BYTES = BP * ROWB, NSL = 3;
constexpr int RP = 16;                                        // padded rank

enum { M_FWD = 0, M_DX = 1, M_G = 2 };

struct LoraArgs {
    // Q operand by LDS-DMA.  FWD: x [M, in], K-major.  DX: dY [M, out], K-major.  G: dY [M, out] as it lies (the reduction runs
    // down its rows: fragments by ds_read_b64_tr_b16).
    const uint16_t *Q;
    int64_t ldq;
    int NQ, K;                    // K: in (FWD), out (DX), M rounded up to 64 (G)
    // P operand.  FWD / DX: generated from W [out, in], mask [out, in], At16 [in, 16], B16 [out, 16].  G: x^T [in, Mp] by DMA.
    const uint16_t *W;
    int64_t ldw;
    const uint8_t *mask;
    const uint16_t *At16, *B16, *A16, *Bt16;
    const uint16_t *Pt;           // G: x [M, in]
    int64_t ldp;
    const uint16_t *zeros;        // G: 128 B of zeros (token rows past M)
    int M;                        // G: tokens
    int NP;                       // out (FWD), in (DX), in (G)
    int out_f, in_f;
    float scaling;
    uint32_t hs;                  // scaling as two fp16 (fast path) or 0
    int sparse, fast;
    // FWD / DX
    uint16_t *Y;
    int64_t ldy;
    const uint16_t *bias;
    // G
    float *part_a, *part_b;       // [nbq][in][16], [nbp][out][16]
    int nbp, nbq, bq;             // bq: rows of Q per tile, 256 or 192
    int dbg;                      // VLMC_LORA_DBG (diagnostics; bits >= 4 give wrong results, only the pace is of interest): 1 full wait before
                                  // the generator, 2 full wait at the top of a step, 4 no generator, 8 no W / mask / slab loads, 16 no Q loads
};

__device__ __forceinline__ int row_off(int row, int ch) { return row * ROWB + ((ch ^ (row & 7)) << 4); }
// [64 k][64 n] image for the transposing reads (as attn_matmul.hip)
__device__ __forceinline__ int tr_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int tr_off(int k, int ch) { return k * ROWB + ((ch ^ tr_swz(k)) << 4); }

template <typename T> __device__ __forceinline__ f32x4_t mfma_x16(const u32x2_t &a, const u32x2_t &b, const f32x4_t &c);
template <> __device__ __forceinline__ f32x4_t mfma_x16<f16_t>(const u32x2_t &a, const u32x2_t &b, const f32x4_t &c) {
    h16x4_t x, y;
    __builtin_memcpy(&x, &a, 8);
    __builtin_memcpy(&y, &b, 8);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma_x16<bf16_t>(const u32x2_t &a, const u32x2_t &b, const f32x4_t &c) {
    s16x4_t x, y;
    __builtin_memcpy(&x, &a, 8);
    __builtin_memcpy(&y, &b, 8);
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ float round16(float v) { return to_f32<T>(from_f32<T>(v)); }

// ---- VMEM the compiler does not see ------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void *gptr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gptr) : "memory", "m0");
}
__device__ __forceinline__ void gload16(u32x4_t &v, const void *p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void gload8(u32x2_t &v, const void *p) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void gload4(uint32_t &v, const void *p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
// the generator's registers of one step: 16 weights, their 16 mask bytes, the step's rank-side fragments
struct GenSet {
    u32x4_t w0, w1, m;
    u32x2_t f[4];
};
template <int N> __device__ __forceinline__ void wait_vm(GenSet &s) {     // ties the registers to the wait
    asm volatile("s_waitcnt vmcnt(%7)"
                 : "+v"(s.w0), "+v"(s.w1), "+v"(s.m), "+v"(s.f[0]), "+v"(s.f[1]), "+v"(s.f[2]), "+v"(s.f[3])
                 : "n"(N)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_plain() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

// ---- one 16 (rows of W) x 64 (columns) piece of W_eff --------------------------------------------------------------------
// Lane (c = lane & 15, g = lane >> 4) holds W[row c][16 g + e], e = 0..15 (w0 | w1) and their mask bytes.  MFMA t = 0..3:
// A operand row rho = 4 g' + r' stands for column 16 g' + 4 t + r', B operand column c for row c, so the result register r
// of lane (c, g) is (B A)[row c][column 16 g + 4 t + r]: element e = 4 t + r of the lane's 16.
template <typename T, bool SPARSE, bool FAST>
__device__ __forceinline__ void gen_piece(const GenSet &s, const u32x2_t (&af)[4], const u32x2_t &bf, const LoraArgs &a, u32x4_t &o0,
                                          u32x4_t &o1) {
    f32x4_t d[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) d[t] = mfma_x16<T>(af[t], bf, f32x4_t{0.f, 0.f, 0.f, 0.f});
    uint32_t out[8];
    const uint32_t w[8] = {s.w0[0], s.w0[1], s.w0[2], s.w0[3], s.w1[0], s.w1[1], s.w1[2], s.w1[3]};
    if constexpr (FAST) {                                                 // packed fp16: every operation is one IEEE rounding
        static_assert(__is_same(T, f16_t), "packed arithmetic exists for fp16 only");
        h16x2_t hs2;
        __builtin_memcpy(&hs2, &a.hs, 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = k >> 1, r0 = (k & 1) * 2;
            h16x2_t dh = {_Float16(d[t][r0]), _Float16(d[t][r0 + 1])};
            dh = dh * hs2;                                                // (s = 1: exact)
            h16x2_t wh, mh;
            __builtin_memcpy(&wh, &w[k], 4);
            const uint32_t mb = (k & 1) ? (s.m[k >> 1] >> 16) : s.m[k >> 1];             // mask bytes 2 k, 2 k + 1 in bits 0, 8
            const uint32_t mbits = ((mb & 1u) | ((mb & 0x100u) << 8)) * 0x3C00u;          // 1.0 / 0.0 per half
            __builtin_memcpy(&mh, &mbits, 4);
            const h16x2_t v = SPARSE ? (wh + dh) * mh : wh * mh + dh;
            __builtin_memcpy(&out[k], &v, 4);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t packed = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * k + h, t = e >> 2, r = e & 3;
                const float d2 = round16<T>(ieee_mul(round16<T>(d[t][r]), a.scaling));
                const float wf = to_f32<T>(uint16_t(w[k] >> (16 * h)));
                const float keep = ((s.m[e >> 2] >> (8 * (e & 3))) & 1u) ? 1.f : 0.f;
                const float v = SPARSE ? ieee_mul(round16<T>(ieee_add(wf, d2)), keep) : round16<T>(ieee_add(ieee_mul(wf, keep), d2));
                packed |= uint32_t(from_f32<T>(v)) << (16 * h);
            }
            out[k] = packed;
        }
    }
    o0 = u32x4_t{out[0], out[1], out[2], out[3]};
    o1 = u32x4_t{out[4], out[5], out[6], out[7]};
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------
// LDS (FWD / DX): Q 2 x 32 KiB | P 3 x 16 KiB | mask 2 x 8 KiB | rank-side slab 3 x 2 KiB = 134 KiB.  EVERY load of the loop
// is an LDS-DMA (no register destinations for the compiler to move): per wave and step 4 pieces of Q (step d + 1), the two
// pieces and the mask of the wave's OWN 16 x 64 piece of W (step d + 2; wave-private: needs no barrier, lands in the P
// image where its W_eff will stand), and 256 B of the step's rank-side fragments (step d + 3; shared: published by the
// barrier after its issuer's wait).  Step d: wait, barrier, issue, first half of the MFMAs, wait for the own piece of step
// d + 1, generate it in place beside the second half of the MFMAs.
// G mode: three slots of [P | Q], both by DMA two steps ahead.
constexpr int M_BYTES = BP * BK, S_BYTES = BK * RP * 2;

template <typename T, int MODE, bool SPARSE, bool FAST, int BQ>
__global__ __launch_bounds__(NTH, 2) void lora_gemm_kernel(const LoraArgs a) {
    constexpr int TQ = BQ / 64, Q_BYTES = BQ * ROWB, SLOT = P_BYTES + Q_BYTES;
    constexpr int QOFF = 0, POFF = 2 * Q_BYTES, MOFF = POFF + 3 * P_BYTES, SOFF = MOFF + 2 * M_BYTES;
    constexpr int LDS_GEN = SOFF + 3 * S_BYTES, LDS_G = NSL * SLOT;
    static_assert(BQ == 256 || BQ == 192, "4 waves x 3 or 4 MFMA tiles of 16 rows");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[MODE == M_G ? LDS_G : LDS_GEN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const uint32_t lds_base = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)lds));

    // tile: contiguous runs of ids per XCD (workgroups b and b + 8 share an L2); q runs fastest: the tiles that share a
    // panel of W are neighbours on one XCD
    int bp, bq;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int qd = nwg >> 3, rm = nwg & 7, xcd = orig & 7;
        const int id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
        bp = id / a.nbq;
        bq = id - bp * a.nbq;
    }
    const int p0 = bp * BP, q0 = bq * BQ;
    const int wp = wave >> 2, wq = wave & 3;
    const int nk = a.K / BK;

    // ---- LDS-DMA pieces of 8 rows x 128 B: Q 3 or 4 per wave and step; G mode: P 2 per wave and step ------------------------
    constexpr int NQP = BQ / 64;
    const uint16_t *srcq[NQP];
    uint32_t dstq[NQP];
    int64_t qstep = BK;                                                   // elements per K-step along a source row / down the source rows
    const uint16_t *srcq_tail[NQP];                                       // G mode: the last step's sources (token rows past M -> a row of zeros)
    if constexpr (MODE == M_G) {
        // dY [M, out] as it lies: 4 sub-images [64 tokens][64 out-features] for the transposing reads
        qstep = int64_t(BK) * a.ldq;
        const int tail_rows = a.M - (nk - 1) * BK;                        // token rows of the last step
#pragma unroll
        for (int v = 0; v < NQP; ++v) {
            const int grp = wave * NQP + v, sub = grp >> 3, r = (grp & 7) * 8 + (lane >> 3), sc = (lane & 7) ^ tr_swz(r);
            const int col = min(q0 + sub * 64, a.out_f - 64) + sc * 8;
            srcq[v] = a.Q + int64_t(r) * a.ldq + col;
            srcq_tail[v] = r < tail_rows ? srcq[v] + int64_t(nk - 1) * qstep : a.zeros + sc * 8;
            dstq[v] = sub * 8192 + (grp & 7) * 1024;
        }
    } else {
#pragma unroll
        for (int v = 0; v < NQP; ++v) {
            const int grp = wave * NQP + v, r = grp * 8 + (lane >> 3), sc = (lane & 7) ^ (r & 7);
            srcq[v] = a.Q + int64_t(min(q0 + r, a.NQ - 1)) * a.ldq + sc * 8;
            srcq_tail[v] = srcq[v];
            dstq[v] = grp * 1024;
        }
    }
    // the wave's own piece of W (or, G mode, its two pieces of x^T) and of the mask
    const uint16_t *srcw[2];
    uint32_t dstw[2];
    int64_t wstep = BK;
    const uint8_t *srcm = nullptr;
    int64_t mstep = BK;
    const uint16_t *srcs = nullptr;                                       // the step's rank-side fragments: 2 KiB, 256 B per wave
    u32x2_t cfrag[4] = {u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}};   // tile-constant rank-side fragments
    int gw0 = 0, gw1 = 0, gm = 0, gs = 0;                                 // where the generator reads / writes inside the slots
    const uint16_t *srcw_tail[2] = {nullptr, nullptr};
    if constexpr (MODE == M_G) {
        // x [M, in] as it lies: 2 sub-images [64 tokens][64 in-features]
        wstep = int64_t(BK) * a.ldp;
        const int tail_rows = a.M - (nk - 1) * BK;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int grp = wave * 2 + v, sub = grp >> 3, r = (grp & 7) * 8 + (lane >> 3), sc = (lane & 7) ^ tr_swz(r);
            const int col = min(p0 + sub * 64, a.in_f - 64) + sc * 8;
            srcw[v] = a.Pt + int64_t(r) * a.ldp + col;
            srcw_tail[v] = r < tail_rows ? srcw[v] + int64_t(nk - 1) * wstep : a.zeros + sc * 8;
            dstw[v] = sub * 8192 + (grp & 7) * 1024;
        }
    } else if constexpr (MODE == M_FWD) {
        // piece = rows p0 + 16 wave .. of W (K-major image rows 16 wave ..), the step's 64 columns
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int r = wave * 16 + v * 8 + (lane >> 3), sc = (lane & 7) ^ (r & 7);
            srcw[v] = a.W + int64_t(min(p0 + r, a.out_f - 1)) * a.ldw + sc * 8;
            dstw[v] = (wave * 16 + v * 8) * ROWB;
        }
        srcm = a.mask + int64_t(min(p0 + wave * 16 + (lane >> 2), a.out_f - 1)) * a.in_f + (lane & 3) * 16;
        srcs = a.At16 + wave * 128 + lane * 8;                            // rows of At16 are the step's columns: 64 x 32 B per step
        const int o = min(p0 + wave * 16 + c, a.out_f - 1);
        cfrag[0] = *reinterpret_cast<const u32x2_t *>(a.B16 + int64_t(o) * RP + 4 * g);
        gw0 = row_off(wave * 16 + c, 2 * g);
        gw1 = row_off(wave * 16 + c, 2 * g + 1);
    } else {
        // piece = rows (K) 16 (wave & 3) .. of the step, columns p0 + 64 (wave >> 2) .. : 16 rows of a [64 k][64 n] image
        const int kb = wave & 3, sh = wave >> 2;
        const int ib = min(p0 + 64 * sh, a.in_f - 64);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int r = kb * 16 + v * 8 + (lane >> 3), sc = (lane & 7) ^ tr_swz(r);
            srcw[v] = a.W + int64_t(r) * a.ldw + ib + sc * 8;
            dstw[v] = sh * 8192 + (kb * 16 + v * 8) * ROWB;
        }
        wstep = int64_t(BK) * a.ldw;
        srcm = a.mask + int64_t(kb * 16 + (lane >> 2)) * a.in_f + ib + (lane & 3) * 16;
        mstep = int64_t(BK) * a.in_f;
        srcs = a.B16 + wave * 128 + lane * 8;                             // rows of B16 are the step's K rows
#pragma unroll
        for (int t = 0; t < 4; ++t)
            cfrag[t] = *reinterpret_cast<const u32x2_t *>(a.At16 + int64_t(ib + 16 * (c >> 2) + 4 * t + (c & 3)) * RP + 4 * g);
        gw0 = sh * 8192 + tr_off(kb * 16 + c, 2 * g);
        gw1 = sh * 8192 + tr_off(kb * 16 + c, 2 * g + 1);
    }
    gm = wave * 1024 + c * 64 + g * 16;
    gs = MODE == M_FWD ? (16 * (c >> 2) + (c & 3)) * 32 + 8 * g : ((wave & 3) * 16 + c) * 32 + 8 * g;      // (+ 4 t rows = 128 t bytes, FWD)

    // the wave's piece of a step: W, mask and fragments out of LDS, W_eff back where W stood
    GenSet gset;
    auto gen_load = [&](int step) {
        const unsigned char *pimg = lds + POFF + (step % 3) * P_BYTES;
        const unsigned char *mimg = lds + MOFF + (step & 1) * M_BYTES, *simg = lds + SOFF + (step % 3) * S_BYTES;
        gset.w0 = *reinterpret_cast<const u32x4_t *>(pimg + gw0);
        gset.w1 = *reinterpret_cast<const u32x4_t *>(pimg + gw1);
        gset.m = *reinterpret_cast<const u32x4_t *>(mimg + gm);
        if constexpr (MODE == M_FWD) {
#pragma unroll
            for (int t = 0; t < 4; ++t) gset.f[t] = *reinterpret_cast<const u32x2_t *>(simg + gs + 128 * t);
        } else {
            gset.f[0] = *reinterpret_cast<const u32x2_t *>(simg + gs);
        }
    };
    auto gen_finish = [&](int step) {
        unsigned char *pimg = lds + POFF + (step % 3) * P_BYTES;
        u32x4_t o0, o1;
        if constexpr (MODE == M_FWD) gen_piece<T, SPARSE, FAST>(gset, gset.f, cfrag[0], a, o0, o1);
        else gen_piece<T, SPARSE, FAST>(gset, cfrag, gset.f[0], a, o0, o1);
        *reinterpret_cast<u32x4_t *>(pimg + gw0) = o0;
        *reinterpret_cast<u32x4_t *>(pimg + gw1) = o1;
    };

    // ---- G mode: the tile's mask dwords (4 consecutive in-features of one out-feature per accumulator fragment) ----------------
    uint32_t mk[TP][TQ];
    if constexpr (MODE == M_G) {
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                const int o = min(q0 + wq * (16 * TQ) + 16 * j + c, a.out_f - 1), ib = min(p0 + wp * 64 + 16 * i + 4 * g, a.in_f - 4);
                mk[i][j] = *reinterpret_cast<const uint32_t *>(a.mask + int64_t(o) * a.in_f + ib);
            }
    }

    f32x4_t acc[TP][TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment offsets inside the images
    const int fq0 = (wq * (16 * TQ) + c) * ROWB + ((g ^ (lane & 7)) << 4), fq1 = (wq * (16 * TQ) + c) * ROWB + (((4 + g) ^ (lane & 7)) << 4);
    const int fp0 = (wp * 64 + c) * ROWB + ((g ^ (lane & 7)) << 4), fp1 = (wp * 64 + c) * ROWB + (((4 + g) ^ (lane & 7)) << 4);
    const int tq = c >> 2, tp = c & 3;                                    // transposing reads: lane 4 q + p of a 16-lane group
    // one K-step of 32: fragments, then TP rows of TQ MFMAs with `between(i)` behind row i (the step's DMA instructions go
    // there: an LDS-DMA costs its wave 60-180 cycles of issue, which the partner wave's MFMAs cover)
    auto multiply = [&](const unsigned char *pimg, const unsigned char *qimg, const int kk, auto &&before, auto &&between) {
        u32x4_t fq[TQ], fp[TP];
        auto tr_frag = [&](const unsigned char *sub_img, int t) {         // 16 columns 16 t .. of a [64 k][64 n] image, k = 32 kk + 8 g ..
            s16x4_t h[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int row = 32 * kk + 8 * g + 4 * half + tq;
                const unsigned char *p = sub_img + tr_off(row, 2 * t + (tp >> 1)) + 8 * (tp & 1);
                h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(const_cast<unsigned char *>(p)));
            }
            u32x4_t f;
            __builtin_memcpy(&f, h, 16);
            return f;
        };
        if constexpr (MODE == M_G) {
#pragma unroll
            for (int j = 0; j < TQ; ++j) fq[j] = tr_frag(qimg + wq * 8192, j);
        } else {
#pragma unroll
            for (int j = 0; j < TQ; ++j) fq[j] = *reinterpret_cast<const u32x4_t *>(qimg + (kk ? fq1 : fq0) + j * 16 * ROWB);
        }
        if constexpr (MODE == M_DX || MODE == M_G) {
#pragma unroll
            for (int i = 0; i < TP; ++i) fp[i] = tr_frag(pimg + wp * 8192, i);
        } else {
#pragma unroll
            for (int i = 0; i < TP; ++i) fp[i] = *reinterpret_cast<const u32x4_t *>(pimg + (kk ? fp1 : fp0) + i * 16 * ROWB);
        }
        before();
#pragma unroll
        for (int i = 0; i < TP; ++i) {
#pragma unroll
            for (int j = 0; j < TQ; ++j) acc[i][j] = mfma16<T>(fp[i], fq[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            between(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto nothing = [] {};

    // ---- the K loop ------------------------------------------------------------------------------------------------------------
    if constexpr (MODE == M_G) {
        constexpr int ND = 2 + NQP;
        auto issue_piece = [&](int step, int v) {                         // v = 0, 1: x; 2 ..: dY
            const uint32_t slot = lds_base + (step % NSL) * SLOT;
            const bool tail = step == nk - 1;
            if (v < 2) glds16(tail ? srcw_tail[v] : srcw[v] + step * wstep, slot + dstw[v]);
            else glds16(tail ? srcq_tail[v - 2] : srcq[v - 2] + step * qstep, slot + P_BYTES + dstq[v - 2]);
        };
        for (int st = 0; st < 2 && st < nk; ++st)
#pragma unroll
            for (int v = 0; v < ND; ++v) issue_piece(st, v);
        for (int d = 0; d < nk; ++d) {
            if (d + 1 < nk) wait_vm_plain<ND>();
            else wait_vm_plain<0>();
            barrier();
            const bool more = d + 2 < nk;
            const unsigned char *img = lds + (d % NSL) * SLOT;
            multiply(img, img + P_BYTES, 0, nothing, [&](int i) { if (more && i < 3) issue_piece(d + 2, i); });
            multiply(img, img + P_BYTES, 1, nothing, [&](int i) { if (more && 3 + i < ND) issue_piece(d + 2, 3 + i); });
        }
    } else {
        // Per step 4 + NQP DMA instructions per wave, in this order: Q(d + 1) x NQP behind the first K-step's MFMA rows, then
        // W(d + 2) x 2, mask(d + 2), slab(d + 3) behind the second's.  Steps past the end are issued all the same with the step
        // index clamped (the last step again, into slots nobody reads any more): the counts of the two waits are constants.
        const int last = nk - 1;
        auto issue_q1 = [&](int t, int v) { glds16(srcq[v] + min(t, last) * BK, lds_base + QOFF + (t & 1) * Q_BYTES + dstq[v]); };
        auto issue_w1 = [&](int t, int v) { glds16(srcw[v] + min(t, last) * wstep, lds_base + POFF + (t % 3) * P_BYTES + dstw[v]); };
        auto issue_m1 = [&](int t) { glds16(srcm + min(t, last) * mstep, lds_base + MOFF + (t & 1) * M_BYTES + wave * 1024); };
        auto issue_slab = [&](int t) {
            if (lane < 16) glds16(srcs + int64_t(min(t, last)) * (BK * RP), lds_base + SOFF + (t % 3) * S_BYTES + wave * 256);
        };
        // prologue: what steps -3, -2 and -1 would have issued
        issue_slab(0);
        issue_w1(0, 0); issue_w1(0, 1); issue_m1(0);
        issue_slab(1);
#pragma unroll
        for (int v = 0; v < NQP; ++v) issue_q1(0, v);
        issue_w1(1, 0); issue_w1(1, 1); issue_m1(1);
        issue_slab(2);
        wait_vm_plain<NQP + 5>();                                         // slab 0 and the own piece of step 0 have landed
        barrier();
        gen_load(0);
        gen_finish(0);
        for (int d = 0; d < nk; ++d) {
            if (a.dbg & 2) wait_vm_plain<0>();
            else wait_vm_plain<4>();                                      // own pieces of Q(d) -- and everything older: slab(d + 1)
            barrier();                                                    // everybody's have; W_eff(d) stands
            const unsigned char *pimg = lds + POFF + (d % 3) * P_BYTES, *qimg = lds + QOFF + (d & 1) * Q_BYTES;
            multiply(pimg, qimg, 0, nothing, [&](int i) { if (i < NQP && !(a.dbg & 16)) issue_q1(d + 1, i); });
            const bool gen = d < last && !(a.dbg & 4);
            multiply(pimg, qimg, 1,
                     [&] {
                         if (gen) {
                             if (a.dbg & 1) wait_vm_plain<0>();
                             else if (!(a.dbg & 32)) wait_vm_plain<NQP + 1>();               // the own piece of step d + 1 (behind it: slab(d + 2), Q(d + 1))
                             gen_load(d + 1);
                         }
                     },
                     [&](int i) {
                         if (!(a.dbg & 8)) {
                             if (i == 0) issue_w1(d + 2, 0);
                             else if (i == 1) issue_w1(d + 2, 1);
                             else if (i == 2) issue_m1(d + 2);
                             else issue_slab(d + 3);
                         }
                         if (i == 1 && gen && !(a.dbg & 64)) gen_finish(d + 1);
                     });
        }
        wait_vm_plain<0>();                                               // (the clamped tail loads)
    }
    barrier();                                                            // every wave is done with the slots, every DMA has landed

    // ---- epilogues -------------------------------------------------------------------------------------------------------------
    if constexpr (MODE != M_G) {
        // Y[q][p]: the wave's 64 x 64 piece through its own 8 KiB of LDS, out as 16 B per lane, 128 B per row
        unsigned char *wl = lds + wave * (16 * TQ * ROWB);
        const uint16_t *bias = a.bias;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            const int p = p0 + wp * 64 + 16 * i + 4 * g;
            if (bias != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = p + r < a.NP ? to_f32<T>(bias[p + r]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                uint16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(bias != nullptr ? acc[i][j][r] + b[r] : acc[i][j][r]);
                const u32x2_t v = {uint32_t(o[0]) | uint32_t(o[1]) << 16, uint32_t(o[2]) | uint32_t(o[3]) << 16};
                const int row = 16 * j + c, chunk = 2 * i + (g >> 1);
                *reinterpret_cast<u32x2_t *>(wl + row * ROWB + ((chunk ^ (row & 7)) << 4) + (g & 1) * 8) = v;
            }
        }
#pragma unroll
        for (int s = 0; s < 2 * TQ; ++s) {
            const int idx = s * 64 + lane, row = idx >> 3, ch = idx & 7;
            const u32x4_t v = *reinterpret_cast<const u32x4_t *>(wl + row * ROWB + ((ch ^ (row & 7)) << 4));
            const int q = q0 + wq * (16 * TQ) + row, p = p0 + wp * 64 + ch * 8;
            if (q < a.NQ && p + 8 <= a.NP) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t *>(a.Y + int64_t(q) * a.ldy + p));
        }
    } else {
        // lane: in-feature i = p0 + 64 wp + 16 a + 4 g + r (rows), out-feature o = q0 + 64 wq + 16 b + c (column)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(mk[0][0]), "+v"(mk[0][1]), "+v"(mk[0][2]), "+v"(mk[0][3]), "+v"(mk[1][0]), "+v"(mk[1][1]), "+v"(mk[1][2]),
                       "+v"(mk[1][3]), "+v"(mk[2][0]), "+v"(mk[2][1]), "+v"(mk[2][2]), "+v"(mk[2][3]), "+v"(mk[3][0]), "+v"(mk[3][1]),
                       "+v"(mk[3][2]), "+v"(mk[3][3])
                     :
                     : "memory");
        unsigned char *wl = lds + wave * 8192;                            // [64 o][64 i] image for the transposing reads
        f32x4_t *red = reinterpret_cast<f32x4_t *>(lds + 65536);          // [wave][8][64 lanes]
        u32x2_t gm[TP][TQ];
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                const bool valid = q0 + wq * (16 * TQ) + 16 * j + c < a.out_f && p0 + wp * 64 + 16 * i + 4 * g < a.in_f;
                uint16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = round16<T>(acc[i][j][r]);                   // the GEMM's output in the 16-bit dtype
                    if (SPARSE && !((mk[i][j] >> (8 * r)) & 1u)) x = 0.f;
                    x = round16<T>(ieee_mul(x, a.scaling));
                    o[r] = valid ? from_f32<T>(x) : uint16_t(0);
                }
                gm[i][j] = u32x2_t{uint32_t(o[0]) | uint32_t(o[1]) << 16, uint32_t(o[2]) | uint32_t(o[3]) << 16};
                *reinterpret_cast<u32x2_t *>(wl + tr_off(16 * j + c, 2 * i + (g >> 1)) + (g & 1) * 8) = gm[i][j];
            }
        // dB^T[j][o] = sum_i A[j][i] Gm[i][o]: the accumulator fragment IS the B operand (k = its row index)
        f32x4_t accb[TQ], acca[TP];
        u32x2_t af[TP], bf[TQ];
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            const int ib = p0 + wp * 64 + 16 * i + 4 * g;
            af[i] = ib < a.in_f ? *reinterpret_cast<const u32x2_t *>(a.A16 + int64_t(c) * a.in_f + ib) : u32x2_t{0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const int ob = q0 + wq * (16 * TQ) + 16 * j + 4 * g;
            bf[j] = ob < a.out_f ? *reinterpret_cast<const u32x2_t *>(a.Bt16 + int64_t(c) * a.out_f + ob) : u32x2_t{0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            accb[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TP; ++i) accb[j] = mfma_x16<T>(af[i], gm[i][j], accb[j]);
        }
        // dA^T[i][j] = sum_o Gm[i][o] B[o][j]: Gm^T through the image (lane c gets in-feature 16 a + c of 4 consecutive o)
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            acca[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                const unsigned char *p = wl + tr_off(16 * j + 4 * g + tq, 2 * i + (tp >> 1)) + 8 * (tp & 1);
                const s16x4_t h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(const_cast<unsigned char *>(p)));
                u32x2_t x;
                __builtin_memcpy(&x, &h, 8);
                acca[i] = mfma_x16<T>(x, bf[j], acca[i]);
            }
        }
        // sums over the waves that share the tile's columns (dB: the two p halves) / rows (dA: the four q quarters)
#pragma unroll
        for (int j = 0; j < TQ; ++j) red[(wave * 8 + j) * 64 + lane] = accb[j];
#pragma unroll
        for (int i = 0; i < TP; ++i) red[(wave * 8 + 4 + i) * 64 + lane] = acca[i];
        __syncthreads();
        if (wp == 0) {
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                const f32x4_t other = red[((wave + 4) * 8 + j) * 64 + lane];
                const int o = q0 + wq * (16 * TQ) + 16 * j + c;
                f32x4_t v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ieee_add(accb[j][r], other[r]);
                if (o < a.out_f) *reinterpret_cast<f32x4_t *>(a.part_b + (int64_t(bp) * a.out_f + o) * RP + 4 * g) = v;
            }
        }
        if (wq == 0) {
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                f32x4_t v = acca[i];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f32x4_t other = red[((wave + w) * 8 + 4 + i) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ieee_add(v[r], other[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ii = p0 + wp * 64 + 16 * i + 4 * g + r;
                    if (ii < a.in_f) a.part_a[(int64_t(bq) * a.in_f + ii) * RP + c] = v[r];
                }
            }
        }
    }
}

// ---- small kernels around it ---------------------------------------------------------------------------------------------------
// 16-bit images of the adapter factors, rounded to the autocast dtype, rank padded to 16 with zeros:
//   At16 [in][16], A16 [16][in], B16 [out][16], Bt16 [16][out]
template <typename T>
__global__ __launch_bounds__(256) void lora_prep_kernel(const float *__restrict__ A, const float *__restrict__ B, int out_f, int in_f, int r,
                                                        uint16_t *__restrict__ At16, uint16_t *__restrict__ A16, uint16_t *__restrict__ B16,
                                                        uint16_t *__restrict__ Bt16, uint16_t *__restrict__ zeros) {
    const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t na = int64_t(in_f) * RP, nb = int64_t(out_f) * RP;
    if (idx < 128) zeros[idx] = 0;
    if (idx < na) {
        const int i = int(idx / RP), j = int(idx % RP);
        const uint16_t v = j < r ? from_f32<T>(A[int64_t(j) * in_f + i]) : uint16_t(0);
        At16[idx] = v;
        A16[int64_t(j) * in_f + i] = v;
    } else if (idx < na + nb) {
        const int64_t e = idx - na;
        const int o = int(e / RP), j = int(e % RP);
        const uint16_t v = j < r ? from_f32<T>(B[int64_t(o) * r + j]) : uint16_t(0);
        B16[e] = v;
        Bt16[int64_t(j) * out_f + o] = v;
    }
}

// dA[j][i] = wd(sum_bq part_a[bq][i][j]),  dB[o][j] = wd(sum_bp part_b[bp][o][j]): tiles in order
template <typename T>
__global__ __launch_bounds__(256) void lora_partial_reduce_kernel(const float *__restrict__ part_a, int nbq, const float *__restrict__ part_b,
                                                                  int nbp, int out_f, int in_f, int r, float *__restrict__ dA,
                                                                  float *__restrict__ dB) {
    const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t na = int64_t(in_f) * RP, nb = int64_t(out_f) * RP;
    const bool is_a = idx < na;
    if (!is_a && idx >= na + nb) return;
    const int64_t e = is_a ? idx : idx - na, n = is_a ? na : nb;
    const int row = int(e / RP), j = int(e % RP), tiles = is_a ? nbq : nbp;
    float *out = is_a ? dA : dB;
    if (j >= r || out == nullptr) return;
    const float *part = (is_a ? part_a : part_b) + e;
    float v = 0.f;
    int t = 0;
    for (; t + 8 <= tiles; t += 8) {                                      // 8 loads in flight, added in tile order
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = part[int64_t(t + u) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u) v = ieee_add(v, x[u]);
    }
    for (; t < tiles; ++t) v = ieee_add(v, part[int64_t(t) * n]);
    if (is_a) dA[int64_t(j) * in_f + row] = round16<T>(v);
    else dB[int64_t(row) * r + j] = round16<T>(v);
}

struct PrepLayout {
    size_t at16, a16, b16, bt16, zeros, total;
};
PrepLayout prep_layout(int64_t out_f, int64_t in_f) {
    PrepLayout l;
    const size_t na = round_up(size_t(in_f) * RP * 2, 256), nb = round_up(size_t(out_f) * RP * 2, 256);
    l.at16 = 0;
    l.a16 = na;
    l.b16 = 2 * na;
    l.bt16 = 2 * na + nb;
    l.zeros = 2 * na + 2 * nb;                    // 256 B of zeros: the token rows past M of the weight-gradient GEMM are read from here
    l.total = l.zeros + 256;
    return l;
}

uint32_t half_bits_if_exact(float s) {                // scaling as fp16 when that loses nothing, else 0
    const _Float16 h = _Float16(s);
    if (float(h) != s) return 0;
    uint16_t b;
    memcpy(&b, &h, 2);
    return uint32_t(b) | uint32_t(b) << 16;
}

int check_common(const char *what, int dtype, int ab_code, int64_t out_f, int64_t in_f, int r, int64_t ldw, const void *W, const void *mask,
                 const void *prep) {
    VLMC_REQUIRE(dtype == VLMC_F16 || dtype == VLMC_BF16, "%s: 16-bit weights only (the fused path; use vlmc_lora_effective_weight otherwise)", what);
    VLMC_REQUIRE(ab_code == (dtype == VLMC_F16 ? 1 : 2), "%s: the autocast dtype must be the weight dtype", what);
    VLMC_REQUIRE(r >= 1 && r <= RP, "%s: rank 1..%d", what, RP);
    VLMC_REQUIRE(out_f > 0 && in_f > 0 && out_f % 64 == 0 && in_f % 64 == 0 && out_f < (int64_t(1) << 30) && in_f < (int64_t(1) << 30),
                 "%s: out_features and in_features must be multiples of 64", what);
    VLMC_REQUIRE(W && mask && prep && aligned16(W) && aligned16(mask) && aligned16(prep) && ldw % 8 == 0 && ldw >= in_f,
                 "%s: W, mask and the prepared factors must be 16-byte aligned, ldw a multiple of 8", what);
    return VLMC_OK;
}

void fill_common(LoraArgs &a, const void *W, int64_t ldw, const uint8_t *mask, const void *prep, int64_t out_f, int64_t in_f, float scaling,
                 int sparse, int dtype) {
    const PrepLayout l = prep_layout(out_f, in_f);
    const char *p = static_cast<const char *>(prep);
    a.W = static_cast<const uint16_t *>(W);
    a.ldw = ldw;
    a.mask = mask;
    a.At16 = reinterpret_cast<const uint16_t *>(p + l.at16);
    a.A16 = reinterpret_cast<const uint16_t *>(p + l.a16);
    a.B16 = reinterpret_cast<const uint16_t *>(p + l.b16);
    a.Bt16 = reinterpret_cast<const uint16_t *>(p + l.bt16);
    a.out_f = int(out_f);
    a.in_f = int(in_f);
    a.scaling = scaling;
    a.sparse = sparse != 0;
    a.hs = dtype == VLMC_F16 ? half_bits_if_exact(scaling) : 0;
    a.fast = dtype == VLMC_F16 && a.hs != 0;
    {
        const char *e = getenv("VLMC_LORA_DBG");
        a.dbg = e ? atoi(e) : 0;
    }
}

// rows of activations per tile: the choice (256 or 192) that leaves the fewer rounds of workgroups on the chip, then the less padding
int pick_bq(int64_t M, int nbp) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    static const int forced = [] {
        const char *e = getenv("VLMC_LORA_BQ");
        return e ? atoi(e) : 0;
    }();
    if (forced == 192 || forced == 256) return forced;
    auto cost = [&](int bq) {
        const int64_t tiles = int64_t(nbp) * ((M + bq - 1) / bq), rounds = (tiles + cus - 1) / cus;
        return double(rounds) * bq;                                       // time ~ rounds x rows per tile
    };
    return cost(192) < cost(256) ? 192 : 256;
}
template <typename T, int MODE, bool SPARSE, int BQ> void launch_q(const LoraArgs &a, hipStream_t s) {
    const unsigned grid = unsigned(a.nbp) * unsigned(a.nbq);
    if constexpr (__is_same(T, f16_t) && MODE != M_G) {
        if (a.fast) {
            VLMC_LAUNCH_TIMED((lora_gemm_kernel<T, MODE, SPARSE, true, BQ>), dim3(grid), dim3(NTH), s, a);
            return;
        }
    }
    VLMC_LAUNCH_TIMED((lora_gemm_kernel<T, MODE, SPARSE, false, BQ>), dim3(grid), dim3(NTH), s, a);
}
template <typename T, int MODE, bool SPARSE> void launch_t(const LoraArgs &a, hipStream_t s) {
    if constexpr (MODE != M_G) {
        if (a.bq == 192) {
            launch_q<T, MODE, SPARSE, 192>(a, s);
            return;
        }
    }
    launch_q<T, MODE, SPARSE, 256>(a, s);
}
template <int MODE> void launch(const LoraArgs &a, int dtype, hipStream_t s) {
    if (dtype == VLMC_F16) {
        if (a.sparse) launch_t<f16_t, MODE, true>(a, s);
        else launch_t<f16_t, MODE, false>(a, s);
    } else {
        if (a.sparse) launch_t<bf16_t, MODE, true>(a, s);
        else launch_t<bf16_t, MODE, false>(a, s);
    }
}

}  // namespace
}  // namespace vlmc

using namespace vlmc;

extern "C" size_t vlmc_sparse_lora_prep_bytes(int64_t out_features, int64_t in_features) {
    if (out_features <= 0 || in_features <= 0) return 0;
    return prep_layout(out_features, in_features).total;
}

extern "C" int vlmc_sparse_lora_prep(const float *A, const float *B, int64_t out_features, int64_t in_features, int r, int ab_code, void *prep,
                                     void *stream) {
    VLMC_REQUIRE(A && B && prep && aligned16(prep), "vlmc_sparse_lora_prep: null or misaligned pointer");
    VLMC_REQUIRE(r >= 1 && r <= RP && out_features > 0 && in_features > 0 && (ab_code == 1 || ab_code == 2),
                 "vlmc_sparse_lora_prep: rank 1..%d, autocast code 1 (fp16) or 2 (bf16)", RP);
    const PrepLayout l = prep_layout(out_features, in_features);
    char *p = static_cast<char *>(prep);
    const int64_t n = (in_features + out_features) * RP;
    const dim3 grid(unsigned((n + 255) / 256));
    uint16_t *at16 = reinterpret_cast<uint16_t *>(p + l.at16), *a16 = reinterpret_cast<uint16_t *>(p + l.a16);
    uint16_t *b16 = reinterpret_cast<uint16_t *>(p + l.b16), *bt16 = reinterpret_cast<uint16_t *>(p + l.bt16);
    if (ab_code == 1)
        hipLaunchKernelGGL(lora_prep_kernel<f16_t>, grid, dim3(256), 0, as_stream(stream), A, B, int(out_features), int(in_features), r, at16, a16, b16, bt16,
                           reinterpret_cast<uint16_t *>(p + l.zeros));
    else
        hipLaunchKernelGGL(lora_prep_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), A, B, int(out_features), int(in_features), r, at16, a16, b16, bt16,
                           reinterpret_cast<uint16_t *>(p + l.zeros));
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparse_lora_prep");
    return VLMC_OK;
}

extern "C" int vlmc_sparse_lora_fwd(const void *X, int64_t M, int64_t ldx, const void *W, int dtype, int64_t out_features, int64_t in_features,
                                    int64_t ldw, const uint8_t *mask, const void *prep, int r, float scaling, int sparse, int ab_code,
                                    const void *bias, void *Y, int64_t ldy, void *stream) {
    if (int rc = check_common("vlmc_sparse_lora_fwd", dtype, ab_code, out_features, in_features, r, ldw, W, mask, prep)) return rc;
    VLMC_REQUIRE(X && Y && aligned16(X) && aligned16(Y) && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= in_features && ldy >= out_features,
                 "vlmc_sparse_lora_fwd: X and Y must be 16-byte aligned with row strides that are multiples of 8");
    VLMC_REQUIRE(M >= 0 && M < (int64_t(1) << 30), "vlmc_sparse_lora_fwd: bad M");
    if (M == 0) return VLMC_OK;
    LoraArgs a{};
    fill_common(a, W, ldw, mask, prep, out_features, in_features, scaling, sparse, dtype);
    a.Q = static_cast<const uint16_t *>(X);
    a.ldq = ldx;
    a.NQ = int(M);
    a.K = int(in_features);
    a.NP = int(out_features);
    a.Y = static_cast<uint16_t *>(Y);
    a.ldy = ldy;
    a.bias = static_cast<const uint16_t *>(bias);
    a.nbp = int((out_features + BP - 1) / BP);
    a.bq = pick_bq(M, a.nbp);
    a.nbq = int((M + a.bq - 1) / a.bq);
    launch<M_FWD>(a, dtype, as_stream(stream));
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparse_lora_fwd");
    return VLMC_OK;
}

extern "C" int vlmc_sparse_lora_bwd_input(const void *dY, int64_t M, int64_t lddy, const void *W, int dtype, int64_t out_features,
                                          int64_t in_features, int64_t ldw, const uint8_t *mask, const void *prep, int r, float scaling,
                                          int sparse, int ab_code, void *dX, int64_t lddx, void *stream) {
    if (int rc = check_common("vlmc_sparse_lora_bwd_input", dtype, ab_code, out_features, in_features, r, ldw, W, mask, prep)) return rc;
    VLMC_REQUIRE(dY && dX && aligned16(dY) && aligned16(dX) && lddy % 8 == 0 && lddx % 8 == 0 && lddy >= out_features && lddx >= in_features,
                 "vlmc_sparse_lora_bwd_input: dY and dX must be 16-byte aligned with row strides that are multiples of 8");
    VLMC_REQUIRE(M >= 0 && M < (int64_t(1) << 30), "vlmc_sparse_lora_bwd_input: bad M");
    if (M == 0) return VLMC_OK;
    LoraArgs a{};
    fill_common(a, W, ldw, mask, prep, out_features, in_features, scaling, sparse, dtype);
    a.Q = static_cast<const uint16_t *>(dY);
    a.ldq = lddy;
    a.NQ = int(M);
    a.K = int(out_features);
    a.NP = int(in_features);
    a.Y = static_cast<uint16_t *>(dX);
    a.ldy = lddx;
    a.bias = nullptr;
    a.nbp = int((in_features + BP - 1) / BP);
    a.bq = pick_bq(M, a.nbp);
    a.nbq = int((M + a.bq - 1) / a.bq);
    launch<M_DX>(a, dtype, as_stream(stream));
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparse_lora_bwd_input");
    return VLMC_OK;
}

namespace {
struct GradWs {
    size_t part_a, part_b, total;
    int nbp, nbq;
};
GradWs grad_ws(int64_t out_f, int64_t in_f) {
    GradWs w;
    w.nbp = int((in_f + BP - 1) / BP);
    w.nbq = int((out_f + 255) / 256);
    w.part_a = 0;
    w.part_b = w.part_a + round_up(size_t(w.nbq) * size_t(in_f) * RP * 4, 256);
    w.total = w.part_b + round_up(size_t(w.nbp) * size_t(out_f) * RP * 4, 256);
    return w;
}
}  // namespace

extern "C" size_t vlmc_sparse_lora_bwd_weight_workspace(int64_t M, int64_t out_features, int64_t in_features) {
    if (M <= 0 || out_features <= 0 || in_features <= 0) return 0;
    return grad_ws(out_features, in_features).total;
}

extern "C" int vlmc_sparse_lora_bwd_weight(const void *dY, int64_t lddy, const void *X, int64_t ldx, int64_t M, int dtype, int64_t out_features,
                                           int64_t in_features, const uint8_t *mask, const void *prep, int r, float scaling, int sparse,
                                           int ab_code, float *dA, float *dB, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_common("vlmc_sparse_lora_bwd_weight", dtype, ab_code, out_features, in_features, r, in_features, dY, mask, prep)) return rc;
    VLMC_REQUIRE(dY && X && aligned16(dY) && aligned16(X) && lddy % 8 == 0 && ldx % 8 == 0 && lddy >= out_features && ldx >= in_features,
                 "vlmc_sparse_lora_bwd_weight: dY and X must be 16-byte aligned with row strides that are multiples of 8");
    VLMC_REQUIRE(M > 0 && M < (int64_t(1) << 24), "vlmc_sparse_lora_bwd_weight: bad M");
    VLMC_REQUIRE(dA || dB, "vlmc_sparse_lora_bwd_weight: nothing to compute");
    const GradWs w = grad_ws(out_features, in_features);
    if (!workspace || workspace_bytes < w.total || (reinterpret_cast<uintptr_t>(workspace) & 255u)) {
        set_error("vlmc_sparse_lora_bwd_weight: a 256-byte aligned workspace of %zu bytes is needed, %zu given", w.total, workspace_bytes);
        return VLMC_EWORKSPACE;
    }
    hipStream_t s = as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    LoraArgs a{};
    fill_common(a, nullptr, in_features, mask, prep, out_features, in_features, scaling, sparse, dtype);
    a.Q = static_cast<const uint16_t *>(dY);
    a.ldq = lddy;
    a.NQ = int(out_features);
    a.M = int(M);
    a.K = int((M + BK - 1) / BK * BK);
    a.Pt = static_cast<const uint16_t *>(X);
    a.ldp = ldx;
    a.NP = int(in_features);
    a.zeros = reinterpret_cast<const uint16_t *>(static_cast<const char *>(prep) + prep_layout(out_features, in_features).zeros);
    a.part_a = reinterpret_cast<float *>(ws + w.part_a);
    a.part_b = reinterpret_cast<float *>(ws + w.part_b);
    a.nbp = w.nbp;
    a.nbq = w.nbq;
    a.bq = 256;
    launch<M_G>(a, dtype, s);
    const int64_t n = (in_features + out_features) * RP;
    if (dtype == VLMC_F16)
        hipLaunchKernelGGL(lora_partial_reduce_kernel<f16_t>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, a.part_a, w.nbq, a.part_b, w.nbp,
                           int(out_features), int(in_features), r, dA, dB);
    else
        hipLaunchKernelGGL(lora_partial_reduce_kernel<bf16_t>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, a.part_a, w.nbq, a.part_b, w.nbp,
                           int(out_features), int(in_features), r, dA, dB);
    VLMC_HIP_CHECK_LAUNCH("vlmc_sparse_lora_bwd_weight");
    return VLMC_OK;
}
