// Fused attention of the calibration forward (gfx950): vlmc_sdpa_fwd.
//
//   O[b][h][q][:] = softmax_k( scale * Q[b][h][q][:] . K[b][h][k][:] ) @ V[b][h][:][:]        16-bit operands, fp32 inside
//
// `F.scaled_dot_product_attention(q, k, v)` -- no mask, no dropout, not causal -- is how current model files write the
// attention the reference's write as `q @ k^T`, softmax, `attn @ v` (eva_vit.py:129-168, modeling_t5.py:520-640); the pruners
// replay every block over the calibration samples (wanda_pruner.py:308-311, :343-346), so it runs 2 x 87 times per prune on
// all 128 samples.  The library kernel behind it takes ~610 us for the ViT-g shape (128 x 16 heads x 257 tokens x 88: 79
// TFLOP/s -- head_dim 88 and 257 tokens fit none of its tiles), 15 % of the GPU time of a whole prune.
//
// Here a head's K and V (257 x 88 x 2 B each) live in LDS for the whole head: no online softmax, one workgroup per (b, h).
//   * S^T = K Q^T, not Q K^T: the accumulator of v_mfma_f32_16x16x32 then holds, per lane, 4 consecutive KEYS of one query --
//     and 4 + 4 of them (two key tiles) are exactly the 8 k-slots the lane must supply as the B operand of O^T = V^T P^T.
//     The probabilities go from the first product's accumulators to the second product's operand IN REGISTERS (the slots of a
//     K-step of 32 keys are keys {4c .. 4c+3} u {16+4c .. 16+4c+3} for lane group c; V^T is fetched with the same slot order).
//   * V lies in LDS as it lies in memory, [key][d]; ds_read_b64_tr_b16 (gfx950's transposing LDS read) hands a lane the 4
//     keys x 1 d-column the A operand wants.  K is read as plain 16-byte fragments.  Rows are 16 B longer than d so that the
//     16 rows of a fragment read start on 16 different bank groups.
//   * a wave owns 32 queries at a time (two accumulator sets share every K / V fragment read: LDS bandwidth is what the
//     kernel is bound by); Q comes straight from global memory into the B operand's layout, once per block of queries.
//   * softmax in fp32 on the accumulators: scale, max and sum over a lane's keys and across the 4 lane groups that share a
//     query (two ds_swizzle-free shuffles), exp2, one division; P is rounded to the operand dtype after normalisation (as
//     the unfused `softmax(...).to(dtype) @ v` does).
// Every (b, h, q) depends on its own Q row and its head's K, V only, through a fixed order of operations: batch-invariant
// like vlmc_linear_fwd / vlmc_attn_matmul (a sample's outputs have the same bits alone or in a group of 128).
#pragma once
#include "common.hpp"
#include "mfma.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>

#ifndef VLMC_SDPA_DBG
#define VLMC_SDPA_DBG 0              // diagnostic builds only (tools/sdpa_ablate.sh): 1 no exp, 2 no P V MFMAs, 4 no K Q MFMAs,
#endif                               // 8 no K / V staging loads, 16 no V fragment reads -- results are garbage, only the pace is of interest

namespace vlmc {

struct SdpaArgs {
    const uint16_t *Q, *K, *V;
    uint16_t *O;
    int64_t sq_b, sq_h, sq_t, sk_b, sk_h, sk_t, sv_b, sv_h, sv_t, so_b, so_h, so_t;     // elements; the d strides are 1
    int H, Tq, Tk, d;
    int qsplit;                 // workgroups per head; workgroup y takes the 32-query blocks y, y + qsplit, ..
    int hpw, nheads;            // heads per workgroup (1, 2 or 4: few queries per head), heads in all
    int dma;                    // K and V rows are 16-byte aligned: staged by LDS-DMA (global_load_lds), no registers in between
    int causal;                 // key j of query i counts iff j <= i (torch's is_causal: the mask is aligned to the top left)
    float scale_log2e;          // scale * log2(e)
};

typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) const uint32_t sdpa_zero_chunk[4] = {0u, 0u, 0u, 0u};      // what padding is "loaded" from

// 64 lanes x 16 B from global memory straight into 1 KiB of LDS at `lds_addr` (wave-uniform), lane l at lds_addr + 16 l
__device__ __forceinline__ void sdpa_glds16(const void *gptr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gptr) : "memory", "m0");
}

struct __attribute__((packed, aligned(2))) SU16x8 { u32x4_t v; };
__device__ __forceinline__ u32x4_t sdpa_load16(const uint16_t *p) { return reinterpret_cast<const SU16x8 *>(p)->v; }
struct __attribute__((packed, aligned(2))) SU16x4 { uint32_t lo, hi; };

// DS = k-steps of 32 along d (d <= 32 DS), the LDS rows hold 32 DS elements + 16 B
// MAXKT: 16-key tiles a head may have -- the accumulators of S^T are registers: 16 (18 for 65 <= d <= 96: the ViT's 257 tokens x
// 88), 8 for heads of at most 128 keys, 4 for at most 64 (the T5 towers: a quarter of the registers, four times the waves per CU).
// FULL: the head has exactly MAXKT key tiles: no run-time guards in the key loops.
// CAUSAL (never together with FULL): keys past the query are masked, key tiles past a block's last query are not computed.
template <typename T, int DS, int MAXKT, bool FULL, bool CAUSAL>
__global__ __launch_bounds__(256) void sdpa_fwd_kernel(const SdpaArgs a) {
    static_assert(!(FULL && CAUSAL), "the causal kernel keeps the run-time guards of the key loops");
    constexpr int DP = 32 * DS, RS = DP * 2 + 16, DT = 2 * DS;                // padded d, row bytes, 16-wide d tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                 // (wave-uniform: LDS-DMA addresses go through m0)
    const int l15 = lane & 15, c = lane >> 4;
    // hpw heads per workgroup: the head's 4 / hpw waves stage its K and V and share its blocks of queries
    const int wph = 4 / a.hpw, sub = wave / wph, hw = wave - sub * wph;         // waves per head, head slot, wave inside the head
    const int bh_raw = blockIdx.x * a.hpw + sub;
    const bool live = bh_raw < a.nheads;                                       // (the last workgroup may have empty slots)
    const int bh = live ? bh_raw : a.nheads - 1, b = bh / a.H, h = bh - b * a.H;
    const int KT = ((a.Tk + 31) >> 5) << 1;                                   // key tiles, even (K-steps of 32 keys)
    const int rows = KT * 16;
    const int image = (rows * RS + 1023) & ~1023;                              // bytes of one K or V image (whole LDS-DMA pieces)
    unsigned char *lk = lds + sub * (2 * image), *lv = lk + image;
    const uint16_t *Kp = a.K + int64_t(b) * a.sk_b + int64_t(h) * a.sk_h;
    const uint16_t *Vp = a.V + int64_t(b) * a.sv_b + int64_t(h) * a.sv_h;
    // ---- the head's K and V into LDS, zero where there is no key / no d ------------------------------------------------
    // LDS-DMA when the rows are 16-byte aligned (they are for d % 8 == 0 views of 16-byte aligned tensors): a wave instruction
    // fills 1 KiB = 64 consecutive 16-byte slots of the image (rows of RS / 16 slots, the last one padding), every slot's
    // lane points at its chunk of K / V or at 16 bytes of zeros; all of a wave's pieces are in flight at once, no register
    // holds anything (staged through registers the 2 x 27 chunks per lane went in four waits: 6 of a head's 23 us).
    if (a.dma) {
        constexpr int SPR = RS / 16;                                           // slots per row
        const int total = image / 16;
        const uint32_t lds_k = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)lk));
        const uint32_t lds_v = uint32_t(uintptr_t((__attribute__((address_space(3))) unsigned char *)lv));
        for (int base = hw * 64; base < total; base += wph * 64) {
            const int j = base + lane, r = j / SPR, ch = j - r * SPR;
            const bool in = !(VLMC_SDPA_DBG & 8) && r < a.Tk && ch * 8 < a.d;     // (rows past the image: r >= rows >= Tk)
            const void *srck = in ? static_cast<const void *>(Kp + int64_t(r) * a.sk_t + ch * 8) : static_cast<const void *>(sdpa_zero_chunk);
            const void *srcv = in ? static_cast<const void *>(Vp + int64_t(r) * a.sv_t + ch * 8) : static_cast<const void *>(sdpa_zero_chunk);
            sdpa_glds16(srck, __builtin_amdgcn_readfirstlane(lds_k + base * 16));
            sdpa_glds16(srcv, __builtin_amdgcn_readfirstlane(lds_v + base * 16));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        constexpr int CPR = DP / 8;                                            // 16-byte chunks per row
        const int total = rows * CPR, nth = 64 * wph, t0 = hw * 64 + lane;
        const u32x4_t zero = {0u, 0u, 0u, 0u};
        for (int base = t0; base < total; base += 8 * nth) {
            u32x4_t kv[8], vv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * nth, r = i / CPR, ch = i - r * CPR;
                const bool in = !(VLMC_SDPA_DBG & 8) && i < total && r < a.Tk && ch * 8 < a.d;
                kv[j] = in ? sdpa_load16(Kp + int64_t(r) * a.sk_t + ch * 8) : zero;
                vv[j] = in ? sdpa_load16(Vp + int64_t(r) * a.sv_t + ch * 8) : zero;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + j * nth, r = i / CPR, ch = i - r * CPR;
                if (i < total) {
                    *reinterpret_cast<u32x4_t *>(lk + r * RS + ch * 16) = kv[j];
                    *reinterpret_cast<u32x4_t *>(lv + r * RS + ch * 16) = vv[j];
                }
            }
        }
    }
    __syncthreads();
    if (!live) return;
    const uint16_t *Qp = a.Q + int64_t(b) * a.sq_b + int64_t(h) * a.sq_h;
    uint16_t *Op = a.O + int64_t(b) * a.so_b + int64_t(h) * a.so_h;
    const int nblk = (a.Tq + 31) >> 5;
    const int tq_ = (lane >> 2) & 3, tp_ = lane & 3;                          // transposing read: row and 8-byte piece inside a group
    const int KTall = FULL ? MAXKT : KT;                                      // (FULL: a compile-time constant, no guards below)
    // One block of 32 queries (TWO) or of at most 16 (the tail of a head).  TWO is a compile-time constant inside: the K loop
    // is straight-line code, fragment reads of the next key tile are in flight during the MFMAs of this one.
    auto block = [&](auto two_c, const int q0, const u32x4_t (&qnow)[2][DS]) {
        constexpr bool TWO = decltype(two_c)::value;
        constexpr int NU = TWO ? 2 : 1;
        // causal: the block's last query is q0 + 31 -- key tiles past it hold nothing but masked keys (whole K-steps of 32 keys)
        const int KTc = CAUSAL ? min(KTall, ((q0 + 32 + 31) >> 5) << 1) : KTall;
        // ---- Q: B operand of S^T, lane (query l15, d chunk c): loaded by the caller one block ahead --------------------------
        u32x4_t fq[NU][DS];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) fq[u][ds] = qnow[u][ds];
        // ---- S^T[key][query]: acc[u][kt][g] = key 16 kt + 4 c + g, query 16 u + l15 --------------------------------------
        f32x4_t acc[NU][MAXKT];
        const unsigned char *kbase = lk + l15 * RS + c * 16;
        u32x4_t fk[2][DS];
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) fk[0][ds] = *reinterpret_cast<const u32x4_t *>(kbase + ds * 64);
#pragma unroll
        for (int kt = 0; kt < MAXKT; ++kt) {
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (!FULL && kt >= KTc) continue;
            if (kt + 1 < MAXKT && (FULL || kt + 1 < KTc)) {
#pragma unroll
                for (int ds = 0; ds < DS; ++ds)
                    fk[(kt + 1) & 1][ds] = *reinterpret_cast<const u32x4_t *>(kbase + (16 * (kt + 1)) * RS + ds * 64);
            }
#pragma unroll
            for (int ds = 0; ds < DS; ++ds)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (VLMC_SDPA_DBG & 4) asm volatile("" : "+v"(acc[u][kt]) : "v"(fk[kt & 1][ds]), "v"(fq[u][ds]));
                    else acc[u][kt] = mfma16<T>(fk[kt & 1][ds], fq[u][ds], acc[u][kt]);
                }
        }
        // ---- softmax over the keys of a query: the lane's 4 KT values, then the 4 lane groups ------------------------------
        // max over the raw scores (the scale is positive), e = exp2(score * k - max * k) in one fma + v_exp, the sum; the
        // probabilities go to the second product UNNORMALISED (e <= 1: the relative rounding is that of e / sum) and the
        // accumulators of O are divided by the sum at the end -- 16 x 6 multiplies instead of 72 x 4.
        // FULL: 16 (MAXKT - 2) key tiles hold real keys whatever Tk is: only the last two are checked for padding.
        float mk[NU], sum[NU], inv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < MAXKT; ++kt) {
                if (!FULL && kt >= KTc) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (CAUSAL) {
                        const int key = 16 * kt + 4 * c + g;
                        acc[u][kt][g] = (key < a.Tk && key <= q0 + 16 * u + l15) ? acc[u][kt][g] : -INFINITY;             // padded and future keys
                    } else if (!FULL || kt >= MAXKT - 2) {
                        acc[u][kt][g] = 16 * kt + 4 * c + g < a.Tk ? acc[u][kt][g] : -INFINITY;                           // padded keys
                    }
                    m = fmaxf(m, acc[u][kt][g]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            mk[u] = -m * a.scale_log2e;
            sum[u] = 0.f;
        }
        // ---- O^T[d][query] = V^T P^T: oacc[u][dt][g] = d 16 dt + 4 c + g, query 16 u + l15 -------------------------------
        // K-step s = key tiles 2 s, 2 s + 1: its probabilities are made (fma, v_exp, rounding: VALU) right before its MFMAs,
        // so that the matrix pipe works on step s while the vector pipe makes step s + 1.
        f32x4_t oacc[NU][DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int u = 0; u < NU; ++u) oacc[u][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // lane group c: keys 32 s + 16 half + 4 c + (0..3), columns 16 dt + (0..15) of V
        const unsigned char *vbase = lv + (4 * c + tq_) * RS + 8 * tp_;
#pragma unroll
        for (int s = 0; s < MAXKT / 2; ++s) {
            if (!FULL && 2 * s >= KTc) continue;
            s16x4_t hv[DT][2];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (VLMC_SDPA_DBG & 16) asm volatile("" : "=v"(hv[dt][half]));
                    else hv[dt][half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(
                        const_cast<unsigned char *>(vbase + (32 * s + 16 * half) * RS + 32 * dt)));
                }
            }
            u32x4_t fp[NU];                                                   // P^T as the B operand of this K-step
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                uint16_t e[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float arg = __builtin_fmaf(acc[u][2 * s + t][g], a.scale_log2e, mk[u]);
                        const float ex = (VLMC_SDPA_DBG & 1) ? arg : __builtin_amdgcn_exp2f(arg);                           // exp2(-inf) = 0
                        sum[u] += ex;
                        e[4 * t + g] = from_f32<T>(ex);
                    }
                __builtin_memcpy(&fp[u], e, 16);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                u32x4_t fv;
                __builtin_memcpy(&fv, hv[dt], 16);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (VLMC_SDPA_DBG & 2) asm volatile("" : "+v"(oacc[u][dt]) : "v"(fv), "v"(fp[u]));
                    else oacc[u][dt] = mfma16<T>(fv, fp[u], oacc[u][dt]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float t = sum[u];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            inv[u] = 1.0f / t;
        }
        // ---- store: lane holds 4 consecutive d of one query -------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q = q0 + 16 * u + l15;
            if (q >= a.Tq) continue;
            uint16_t *orow = Op + int64_t(q) * a.so_t;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int e0 = 16 * dt + 4 * c;
                if (e0 >= a.d) continue;
                uint16_t e[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) e[g] = from_f32<T>(oacc[u][dt][g] * inv[u]);
                SU16x4 v;                                                     // (d is a multiple of 8: e0 < d means e0 + 3 < d)
                __builtin_memcpy(&v, e, 8);
                *reinterpret_cast<SU16x4 *>(orow + e0) = v;
            }
        }
    };
    auto load_q = [&](const int q0, u32x4_t (&dst)[2][DS]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = min(q0 + 16 * u + l15, a.Tq - 1);                   // (rows past the end repeat the last: never stored)
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                const int e = 32 * ds + 8 * c;
                dst[u][ds] = e < a.d ? sdpa_load16(Qp + int64_t(q) * a.sq_t + e) : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    };
    const int step = wph * a.qsplit;
    int blk = blockIdx.y * wph + hw;
    u32x4_t qa[2][DS], qb[2][DS];
    if (blk < nblk) load_q(blk * 32, qa);
    for (; blk < nblk; blk += step) {
        const int q0 = blk * 32;
        if (blk + step < nblk) load_q((blk + step) * 32, qb);                 // the next block's queries: in flight during this block
        if (q0 + 16 < a.Tq) block(std::true_type{}, q0, qa);                  // (wave-uniform)
        else block(std::false_type{}, q0, qa);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) qa[u][ds] = qb[u][ds];
    }
}

// keys a head may have for a given head_dim (the accumulators of S^T live in registers, K and V in LDS)
static int sdpa_max_keys(int d) {
    const int ds = (d + 31) / 32;
    return ds == 4 ? 256 : 288;          // (head_dim <= 96: 288 keys -- the ViT's 257 tokens, and the Q-Former's cross-attention to them at head_dim 64)
}

template <typename T, int DS, int MAXKT, bool FULL, bool CAUSAL> static int sdpa_launch2(const SdpaArgs &a, int64_t bh, size_t lds, hipStream_t s) {
    static PerDeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sdpa_fwd_kernel<T, DS, MAXKT, FULL, CAUSAL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            set_error("vlmc_sdpa_fwd: cannot reserve LDS");
            return VLMC_EHIP;
        }
        once.mark(dev);
    }
    const dim3 grid{unsigned((bh + a.hpw - 1) / a.hpw), unsigned(a.qsplit)}, block{256};
    const LaunchEvents ev = take_launch_events();
    if (ev.start || ev.stop) hipExtLaunchKernelGGL((sdpa_fwd_kernel<T, DS, MAXKT, FULL, CAUSAL>), grid, block, lds, s, ev.start, ev.stop, 0, a);
    else hipLaunchKernelGGL((sdpa_fwd_kernel<T, DS, MAXKT, FULL, CAUSAL>), grid, block, lds, s, a);
    return VLMC_OK;
}

template <typename T, int DS, int MAXKT> static int sdpa_launch1(const SdpaArgs &a, int64_t bh, size_t lds, int KT, hipStream_t s) {
    if (a.causal) return sdpa_launch2<T, DS, MAXKT, false, true>(a, bh, lds, s);
    return KT == MAXKT ? sdpa_launch2<T, DS, MAXKT, true, false>(a, bh, lds, s) : sdpa_launch2<T, DS, MAXKT, false, false>(a, bh, lds, s);
}

template <typename T, int DS> static int sdpa_launch(const SdpaArgs &a, int64_t bh, hipStream_t s) {
    constexpr int RS = 32 * DS * 2 + 16, BIG = DS == 4 ? 16 : 18;
    const int KT = ((a.Tk + 31) >> 5) << 1;
    const size_t lds = size_t(a.hpw) * 2 * ((size_t(KT) * 16 * RS + 1023) & ~size_t(1023));
    if (KT <= 4) return sdpa_launch1<T, DS, 4>(a, bh, lds, KT, s);
    if (KT <= 8) return sdpa_launch1<T, DS, 8>(a, bh, lds, KT, s);         // (a decoder-only tower's 65 .. 128 tokens)
    return sdpa_launch1<T, DS, BIG>(a, bh, lds, KT, s);
}


// one translation unit per dtype (sdpa.hip: fp16 + the C entry point, sdpa_bf16.hip: bf16) so that `make -j` compiles them side by side
int sdpa_dispatch_f16(const SdpaArgs &a, int64_t bh, int ds, hipStream_t s);
int sdpa_dispatch_bf16(const SdpaArgs &a, int64_t bh, int ds, hipStream_t s);

template <typename T> static int sdpa_dispatch(const SdpaArgs &a, int64_t bh, int ds, hipStream_t s) {
    // (head_dim <= 32 runs in the 64-wide instantiation: zero-padded columns, one instantiation fewer to compile)
    return ds <= 2 ? sdpa_launch<T, 2>(a, bh, s) : ds == 3 ? sdpa_launch<T, 3>(a, bh, s) : sdpa_launch<T, 4>(a, bh, s);
}

}  // namespace vlmc
