// The attention of a replayed block, WRITTEN AS THE REFERENCE WRITES IT, in one launch (gfx950): vlmc_attn_fwd.
//
// The reference's model files spell attention out as separate tensor ops, every one of them rounding to the 16-bit dtype
// (eva_vit.py:145-164, modeling_t5.py:588-640, Qformer.py:205-246, modeling_llama.py):
//     scores = q @ k^T                      -> wd( sum_d q k )                      [vlmc_attn_matmul's bits]
//     scores = scores / sqrt(d)             -> wd( float(scores) * c )              (torch multiplies by the fp32 reciprocal)
//     scores += position_bias (+ mask)      -> wd( float(scores) + float(bias) )    up to two addends, broadcast over b / h / q
//     probs  = softmax(scores.float(), -1).type_as(scores)   or   scores.softmax(-1)    -> wd( softmax32(float(scores)) )
//     out    = probs @ v                    -> wd( sum_k probs v )                  [vlmc_attn_matmul's bits]
// Unfused that is 5-7 launches and ~8 passes over the [B, H, Tq, Tk] scores in HBM (270 MB per ViT-g block and pass).  Here a
// head's K and V live in LDS, the scores of 16 queries live in the accumulators of S^T = K Q^T, and every rounding above is
// applied in registers in the same place: the output has the BITS of the unfused sequence run on this library's kernels
// (vlmc_attn_matmul, torch's elementwise ops, vlmc_softmax_rows) -- tests/test_attn_fused_gpu.py compares them, and
// vlmc/forward.py compares them once more at run time before it trusts a signature.
//   * both products take their K-steps of 32 in ascending order with the slots in natural order, like attn_matmul.hip (the
//     matrix core's sum depends on the slot order: tools/micro/mfma_perm.hip), so the probabilities of a K-step go through a
//     1 KB per-wave LDS scratch to reach the B operand's layout (two 8-byte writes, one 16-byte read per lane);
//   * the softmax is the canonical order of softmax_order.hpp: class sums per accumulator register, two in-register levels,
//     two cross-lane ones;
//   * V lies in LDS as in memory and is read with ds_read_b64_tr_b16; K and V are staged by LDS-DMA when their rows are
//     16-byte aligned;
//   * the output is written as [B, Tq, H, d]: the model's `.transpose(1, 2).reshape(B, Tq, H d)` is then a view.
// Batch- and padding-invariant: (b, h, q) depends on its own row of Q, its head's K and V and its row of the addends, through
// a fixed order; keys masked at the dtype's minimum behind a row's live keys contribute exact zeros.
#pragma once
#include "common.hpp"
#include "mfma.hpp"
#include "softmax_order.hpp"

#include <cmath>
#include <cstdlib>

namespace vlmc {

struct AttnArgs {
    const uint16_t *Q, *K, *V;
    uint16_t *O;
    const uint16_t *B0, *B1;                                                     // addends (nullable)
    int64_t sq_b, sq_h, sq_t, sk_b, sk_h, sk_t, sv_b, sv_h, sv_t, so_b, so_h, so_t;  // elements; the d strides are 1
    int64_t s0_b, s0_h, s0_q, s0_k, s1_b, s1_h, s1_q, s1_k;                       // 0 = broadcast (batch, head, query); key stride >= 1
    int H, Tq, Tk, d;
    int nblk, bpw;              // 16-query blocks of a head; blocks per wave (workgroup y takes blocks y NW bpw .. of the head)
    int dma;                    // K and V rows are 16-byte aligned: staged by LDS-DMA
    int has_mul;
    float mul;
    // a PADDED group of ragged samples (vlmc_attn_fwd_lens), either may be NULL: klen[b] = the keys of batch entry b that are real -- the
    // caller vouches that every key behind them is masked out by an addend (probability exactly 0): their tiles are neither staged nor
    // multiplied; qlen[b] = its real queries: the rows behind them are written as zeros, not computed
    const int32_t *qlen, *klen;
};

typedef short af_s16x4_t __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) const uint32_t attn_zero_chunk[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void attn_glds16(const void *gptr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gptr) : "memory", "m0");
}

struct __attribute__((packed, aligned(2))) AU16x8 { u32x4_t v; };
__device__ __forceinline__ u32x4_t attn_load16(const uint16_t *p) { return reinterpret_cast<const AU16x8 *>(p)->v; }
struct __attribute__((packed, aligned(2))) AU16x4 { uint32_t lo, hi; };

constexpr int kAttnScratchRow = 80;                        // bytes per query row of the P scratch: 32 keys + 16 B (bank spread)
constexpr int kAttnScratch = 16 * kAttnScratchRow;         // per wave

// the bits of one rounding to the dtype, as an fp32 value again
template <typename T> __device__ __forceinline__ float attn_round(float v) { return to_f32<T>(from_f32<T>(v)); }

// 4 consecutive addend entries (keys key0 .. key0 + 3 of one query row), zeros past the row's end
__device__ __forceinline__ AU16x4 attn_bias4(const uint16_t *row, int key0, int Tk, int64_t sk) {
    AU16x4 r{0u, 0u};
    if (sk == 1 && key0 + 3 < Tk) return *reinterpret_cast<const AU16x4 *>(row + key0);
    uint16_t e[4] = {0, 0, 0, 0};                                         // (a permuted bias -- T5's [T, T, H] table lookup -- or the row's end)
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (key0 + j < Tk) e[j] = row[int64_t(key0 + j) * sk];
    __builtin_memcpy(&r, e, 8);
    return r;
}

// DS = K-steps of 32 along d; MAXKT = 16-key tiles a head may have (the scores of a block of queries are registers); NW = waves
// per workgroup: 8 where a head's K and V leave room for one workgroup per CU only (two waves per SIMD: one wave's softmax runs
// on the vector pipe while the other's products run on the matrix pipe)
// NADD = addends (0, 1, 2): their prefetched entries are registers.  MINW = waves per SIMD the register budget is held to.
template <typename T, int DS, int MAXKT, int NW, int NADD, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void attn_fused_kernel(const AttnArgs a) {
    constexpr int DP = 32 * DS, RS = DP * 2 + 16, DT = 2 * DS, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, c = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int image = ((((a.Tk + 31) >> 5) << 5) * RS + 1023) & ~1023;         // (the layout is the launch's: lengths only shorten the loops)
    const int Tk = a.klen != nullptr ? max(1, min(a.Tk, a.klen[b])) : a.Tk;  // this batch entry's live keys
    const int Tql = a.qlen != nullptr ? max(0, min(a.Tq, a.qlen[b])) : a.Tq; // .. and live queries
    const int KT = ((Tk + 31) >> 5) << 1;                                     // key tiles, even (K-steps of 32 keys)
    const int rows = KT * 16;
    unsigned char *lk = lds, *lv = lds + image, *lp = lds + 2 * image + wave * kAttnScratch;
    const uint16_t *Kp = a.K + int64_t(b) * a.sk_b + int64_t(h) * a.sk_h;
    const uint16_t *Vp = a.V + int64_t(b) * a.sv_b + int64_t(h) * a.sv_h;
    // ---- the head's K and V into LDS, zero where there is no key / no d -------------------------------------------------
    if (a.dma) {
        constexpr int SPR = RS / 16;
        const int total = (rows * RS + 1023) / 1024 * 64;                      // 16-byte pieces, whole wave-instructions
        const uint32_t lds_k = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)lk));
        const uint32_t lds_v = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)lv));
        for (int base = wave * 64; base < total; base += NT) {
            const int j = base + lane, r = j / SPR, ch = j - r * SPR;
            const bool in = r < Tk && ch * 8 < a.d;
            const void *srck = in ? static_cast<const void *>(Kp + int64_t(r) * a.sk_t + ch * 8) : static_cast<const void *>(attn_zero_chunk);
            const void *srcv = in ? static_cast<const void *>(Vp + int64_t(r) * a.sv_t + ch * 8) : static_cast<const void *>(attn_zero_chunk);
            attn_glds16(srck, __builtin_amdgcn_readfirstlane(lds_k + base * 16));
            attn_glds16(srcv, __builtin_amdgcn_readfirstlane(lds_v + base * 16));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        constexpr int CPR = DP / 8;
        const int total = rows * CPR;
        const u32x4_t zero = {0u, 0u, 0u, 0u};
        for (int base = tid; base < total; base += 8 * NT) {
            u32x4_t kv[8], vv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * NT, r = i / CPR, ch = i - r * CPR;
                const bool in = i < total && r < Tk && ch * 8 < a.d;
                kv[j] = in ? attn_load16(Kp + int64_t(r) * a.sk_t + ch * 8) : zero;
                vv[j] = in ? attn_load16(Vp + int64_t(r) * a.sv_t + ch * 8) : zero;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * NT, r = i / CPR, ch = i - r * CPR;
                if (i < total) {
                    *reinterpret_cast<u32x4_t *>(lk + r * RS + ch * 16) = kv[j];
                    *reinterpret_cast<u32x4_t *>(lv + r * RS + ch * 16) = vv[j];
                }
            }
        }
    }
    __syncthreads();
    const uint16_t *Qp = a.Q + int64_t(b) * a.sq_b + int64_t(h) * a.sq_h;
    uint16_t *Op = a.O + int64_t(b) * a.so_b + int64_t(h) * a.so_h;
    const uint16_t *B0p = NADD >= 1 ? a.B0 + int64_t(b) * a.s0_b + int64_t(h) * a.s0_h : nullptr;
    const uint16_t *B1p = NADD >= 2 ? a.B1 + int64_t(b) * a.s1_b + int64_t(h) * a.s1_h : nullptr;
    const int nblk = (a.Tq + 15) >> 4;
    const int tq_ = (lane >> 2) & 3, tp_ = lane & 3;                          // transposing read: row and 8-byte piece inside a group
    const float ninf = -__builtin_inff();

    for (int i = 0; i < a.bpw; ++i) {
        const int blk = (blockIdx.y * a.bpw + i) * NW + wave;
        if (blk >= nblk) break;
        const int q0 = blk * 16;
        const int q = q0 + l15;
        const bool qlive = q < a.Tq;
        if (q0 >= Tql) {                                                      // a block of padding queries: zeros, nothing computed
            if (qlive) {
                uint16_t *orow = Op + int64_t(q) * a.so_t;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int e0 = 16 * dt + 4 * c;
                    if (e0 < a.d) *reinterpret_cast<AU16x4 *>(orow + e0) = AU16x4{0u, 0u};
                }
            }
            continue;
        }
        const int qc = qlive ? q : a.Tq - 1;                                  // (rows past the end repeat the last: never stored)
        // ---- Q: B operand of S^T, lane (query l15, d chunk c) ------------------------------------------------------------
        u32x4_t fq[DS];
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            const int e = 32 * ds + 8 * c;
            fq[ds] = e < a.d ? attn_load16(Qp + int64_t(qc) * a.sq_t + e) : u32x4_t{0u, 0u, 0u, 0u};
        }
        // ---- the addends of this block of queries: requested now, used after the first product ---------------------------
        AU16x4 b0[NADD >= 1 ? MAXKT : 1], b1[NADD >= 2 ? MAXKT : 1];
        if constexpr (NADD >= 1) {
            const uint16_t *row = B0p + int64_t(qc) * a.s0_q;
#pragma unroll
            for (int kt = 0; kt < MAXKT; ++kt)
                if (kt < KT) b0[kt] = attn_bias4(row, 16 * kt + 4 * c, a.Tk, a.s0_k);
        }
        if constexpr (NADD >= 2) {
            const uint16_t *row = B1p + int64_t(qc) * a.s1_q;
#pragma unroll
            for (int kt = 0; kt < MAXKT; ++kt)
                if (kt < KT) b1[kt] = attn_bias4(row, 16 * kt + 4 * c, a.Tk, a.s1_k);
        }
        // ---- S^T[key][query]: acc[kt][g] = key 16 kt + 4 c + g, query l15; K-steps of 32 along d in ascending order ------
        f32x4_t acc[MAXKT];
        const unsigned char *kbase = lk + l15 * RS + c * 16;
        u32x4_t fk[2][DS];                                                    // the next tile's fragments are in flight during this tile's MFMAs
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) fk[0][ds] = *reinterpret_cast<const u32x4_t *>(kbase + ds * 64);
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            acc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (kt >= KT) continue;
            if (kt + 1 < MAXKT && kt + 1 < KT) {
#pragma unroll
                for (int ds = 0; ds < DS; ++ds)
                    fk[(kt + 1) & 1][ds] = *reinterpret_cast<const u32x4_t *>(kbase + (16 * (kt + 1)) * RS + ds * 64);
            }
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) acc[kt] = mfma16<T>(fk[kt & 1][ds], fq[ds], acc[kt]);
        }
        // ---- the elementwise chain, every step rounded to the dtype like the tensor op it stands for ---------------------
        float m = ninf;
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            if (kt >= KT) continue;
            uint16_t e0[4], e1[4];
            if constexpr (NADD >= 1) __builtin_memcpy(e0, &b0[kt], 8);
            if constexpr (NADD >= 2) __builtin_memcpy(e1, &b1[kt], 8);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = attn_round<T>(acc[kt][g]);
                if (a.has_mul) {
                    float t = ieee_mul(s, a.mul);
                    asm volatile("" : "+v"(t));                              // (a product of its own: no fused multiply + convert)
                    s = attn_round<T>(t);
                }
                if constexpr (NADD >= 1) {
                    float t = ieee_add(s, to_f32<T>(e0[g]));
                    asm volatile("" : "+v"(t));                              // (an fp32 sum of its own, rounded to the dtype afterwards)
                    s = attn_round<T>(t);
                }
                if constexpr (NADD >= 2) {
                    float t = ieee_add(s, to_f32<T>(e1[g]));
                    asm volatile("" : "+v"(t));
                    s = attn_round<T>(t);
                }
                s = 16 * kt + 4 * c + g < Tk ? s : ninf;                      // keys of the padding
                acc[kt][g] = s;
                m = fmaxf(m, s);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, kWave));
        m = fmaxf(m, __shfl_xor(m, 32, kWave));
        // ---- softmax in the canonical order: class 4 c + g sums its keys over the tiles in ascending order ----------------
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
            if (kt >= KT) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float e = softmax_exp(acc[kt][g], m);
                acc[kt][g] = e;
                cs[g] = ieee_add(cs[g], e);
            }
        }
        float tot = ieee_add(ieee_add(cs[0], cs[1]), ieee_add(cs[2], cs[3]));   // classes c ^ 1, then c ^ 2
        tot = ieee_add(tot, __shfl_xor(tot, 16, kWave));                         // c ^ 4
        tot = ieee_add(tot, __shfl_xor(tot, 32, kWave));                         // c ^ 8
        const float inv = softmax_inv(tot);
        // ---- O^T[d][query] = V^T P^T: K-steps of 32 keys in ascending order, natural slot order ---------------------------
        f32x4_t oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const unsigned char *vbase = lv + (8 * c + tq_) * RS + 8 * tp_;
        unsigned char *pw = lp + l15 * kAttnScratchRow + 8 * c;                  // my 4 keys of a tile: row = query, + 32 B for the odd tile
        const unsigned char *pr = lp + l15 * kAttnScratchRow + 16 * c;           // my 8 slots of the K-step
        u32x4_t fp[2];
        // P of a K-step: rounded, through the wave's scratch into the B operand's layout (same wave, LDS in order: the writes are
        // visible to the read; the previous step's read was issued before these writes).  The next step's P is made while this
        // step's products run.
        auto make_p = [&](const f32x4_t &t0, const f32x4_t &t1) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint16_t e[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) e[g] = from_f32<T>(softmax_prob(t ? t1[g] : t0[g], inv));
                u32x2_t w;
                __builtin_memcpy(&w, e, 8);
                *reinterpret_cast<u32x2_t *>(pw + 32 * t) = w;
            }
            return *reinterpret_cast<const u32x4_t *>(pr);
        };
        fp[0] = make_p(acc[0], acc[1]);
#pragma unroll
        for (int s = 0; s < MAXKT / 2; ++s) {
            if (2 * s >= KT) continue;
            af_s16x4_t hv[DT][2];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    hv[dt][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) af_s16x4_t *)(
                        const_cast<unsigned char *>(vbase + (32 * s + 4 * half) * RS + 32 * dt)));
            if (s + 1 < MAXKT / 2 && 2 * (s + 1) < KT) fp[(s + 1) & 1] = make_p(acc[2 * s + 2 < MAXKT ? 2 * s + 2 : 0], acc[2 * s + 3 < MAXKT ? 2 * s + 3 : 0]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                u32x4_t fv;
                __builtin_memcpy(&fv, hv[dt], 16);
                oacc[dt] = mfma16<T>(fv, fp[s & 1], oacc[dt]);
            }
        }
        // ---- store: lane holds 4 consecutive d of one query ---------------------------------------------------------------
        if (qlive) {
            uint16_t *orow = Op + int64_t(q) * a.so_t;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int e0 = 16 * dt + 4 * c;
                if (e0 >= a.d) continue;
                uint16_t e[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) e[g] = q < Tql ? from_f32<T>(oacc[dt][g]) : uint16_t(0);     // (padding queries of a live block: zeros)
                AU16x4 v;
                __builtin_memcpy(&v, e, 8);
                *reinterpret_cast<AU16x4 *>(orow + e0) = v;
            }
        }
    }
}

// keys a head may have for a given head_dim: K and V images + the waves' scratch within 160 KB of LDS, at most 512
static int attn_max_keys(int d) {
    const int ds = d <= 64 ? 2 : (d + 31) / 32;
    const size_t rs = size_t(64) * ds + 16;
    size_t k = (size_t(160) * 1024 - (ds == 4 ? 4 : 8) * kAttnScratch - 1024) / (2 * rs);
    k = k / 32 * 32;
    return int(k > 512 ? 512 : k);
}

template <typename T, int DS, int MAXKT, int NW, int NADD, int MINW> static int attn_launch2(const AttnArgs &a, int64_t bh, hipStream_t s) {
    constexpr int RS = 32 * DS * 2 + 16;
    const int KT = ((a.Tk + 31) >> 5) << 1;
    const size_t lds = 2 * ((size_t(KT) * 16 * RS + 1023) & ~size_t(1023)) + NW * kAttnScratch;
    static PerDeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fused_kernel<T, DS, MAXKT, NW, NADD, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            set_error("vlmc_attn_fwd: cannot reserve LDS");
            return VLMC_EHIP;
        }
        once.mark(dev);
    }
    const dim3 grid{unsigned(bh), unsigned((a.nblk + NW * a.bpw - 1) / (NW * a.bpw))}, block{64 * NW};
    const LaunchEvents ev = take_launch_events();
    if (ev.start || ev.stop) hipExtLaunchKernelGGL((attn_fused_kernel<T, DS, MAXKT, NW, NADD, MINW>), grid, block, lds, s, ev.start, ev.stop, 0, a);
    else hipLaunchKernelGGL((attn_fused_kernel<T, DS, MAXKT, NW, NADD, MINW>), grid, block, lds, s, a);
    return VLMC_OK;
}

template <typename T, int DS, int NADD> static int attn_launch(const AttnArgs &a, int64_t bh, hipStream_t s) {
    const int KT = ((a.Tk + 31) >> 5) << 1;
    if (KT <= 4) return attn_launch2<T, DS, 4, 4, NADD, 3>(a, bh, s);
    if (KT <= 8) return attn_launch2<T, DS, 8, 4, NADD, 3>(a, bh, s);
    if (KT <= 12) return attn_launch2<T, DS, 12, 4, NADD, 2>(a, bh, s);
    if constexpr (DS == 4) return attn_launch2<T, DS, 18, 4, NADD, 1>(a, bh, s);  // (head_dim 128: at most 288 keys fit, with four waves' scratch)
    else if (KT <= 18) return attn_launch2<T, DS, 18, 8, NADD, 2>(a, bh, s);
    else if constexpr (DS == 2) return attn_launch2<T, DS, 32, 8, NADD, 2>(a, bh, s);
    else return attn_launch2<T, DS, 22, 8, NADD, 2>(a, bh, s);                    // (head_dim 96: at most 352 keys fit)
}

// one translation unit per (dtype, number of addends): `make -j` compiles them side by side
template <typename T, int NADD> static int attn_dispatch(const AttnArgs &a, int64_t bh, int ds, hipStream_t s) {
    return ds <= 2 ? attn_launch<T, 2, NADD>(a, bh, s) : ds == 3 ? attn_launch<T, 3, NADD>(a, bh, s) : attn_launch<T, 4, NADD>(a, bh, s);
}
int attn_dispatch_f16_0(const AttnArgs &a, int64_t bh, int ds, hipStream_t s);
int attn_dispatch_f16_1(const AttnArgs &a, int64_t bh, int ds, hipStream_t s);
int attn_dispatch_f16_2(const AttnArgs &a, int64_t bh, int ds, hipStream_t s);
int attn_dispatch_bf16_0(const AttnArgs &a, int64_t bh, int ds, hipStream_t s);
int attn_dispatch_bf16_1(const AttnArgs &a, int64_t bh, int ds, hipStream_t s);
int attn_dispatch_bf16_2(const AttnArgs &a, int64_t bh, int ds, hipStream_t s);

}  // namespace vlmc
