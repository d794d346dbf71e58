// vlmc_sdpa_fwd: the C entry point and the fp16 instantiations of csrc/sdpa_kernel.hpp (the kernel and its description live there;
// sdpa_bf16.hip holds the bf16 instantiations).
#include "sdpa_kernel.hpp"

namespace vlmc {
int sdpa_dispatch_f16(const SdpaArgs &a, int64_t bh, int ds, hipStream_t s) { return sdpa_dispatch<f16_t>(a, bh, ds, s); }
}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_sdpa_max_keys(int64_t head_dim) {
    if (head_dim <= 0 || head_dim > 128 || head_dim % 8 != 0) return 0;
    return sdpa_max_keys(int(head_dim));
}

extern "C" int vlmc_sdpa_fwd(const void *Q, const void *K, const void *V, void *O, int dtype, int64_t batch, int64_t heads,
                             int64_t Tq, int64_t Tk, int64_t head_dim, int64_t sq_b, int64_t sq_h, int64_t sq_t, int64_t sk_b,
                             int64_t sk_h, int64_t sk_t, int64_t sv_b, int64_t sv_h, int64_t sv_t, int64_t so_b, int64_t so_h,
                             int64_t so_t, float scale, int causal, void *stream) {
    VLMC_REQUIRE(dtype == VLMC_F16 || dtype == VLMC_BF16, "vlmc_sdpa_fwd: dtype must be VLMC_F16 or VLMC_BF16");
    VLMC_REQUIRE(causal == 0 || causal == 1, "vlmc_sdpa_fwd: causal must be 0 or 1");
    VLMC_REQUIRE(Q && K && V && O, "vlmc_sdpa_fwd: null pointer");
    VLMC_REQUIRE(batch > 0 && heads > 0 && Tq > 0 && Tk > 0, "vlmc_sdpa_fwd: empty problem");
    VLMC_REQUIRE(head_dim > 0 && head_dim <= 128 && head_dim % 8 == 0, "vlmc_sdpa_fwd: head_dim must be a multiple of 8, at most 128 (got %lld)",
                 (long long)head_dim);
    VLMC_REQUIRE(Tk <= sdpa_max_keys(int(head_dim)), "vlmc_sdpa_fwd: %lld keys per head, at most %d for head_dim %lld (vlmc_sdpa_max_keys)",
                 (long long)Tk, sdpa_max_keys(int(head_dim)), (long long)head_dim);
    VLMC_REQUIRE(batch * heads < (int64_t(1) << 31) && Tq < (int64_t(1) << 24) && heads < (int64_t(1) << 20), "vlmc_sdpa_fwd: shape too large");
    VLMC_REQUIRE(sq_t >= 0 && sk_t >= 0 && sv_t >= 0 && so_t >= head_dim, "vlmc_sdpa_fwd: bad row strides");
    VLMC_REQUIRE(((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) |
                   reinterpret_cast<uintptr_t>(O)) & 1u) == 0, "vlmc_sdpa_fwd: pointers must be 2-byte aligned");
    VLMC_REQUIRE(std::isfinite(scale) && scale > 0.f, "vlmc_sdpa_fwd: scale must be finite and positive");
    SdpaArgs a{};
    a.Q = static_cast<const uint16_t *>(Q), a.K = static_cast<const uint16_t *>(K), a.V = static_cast<const uint16_t *>(V);
    a.O = static_cast<uint16_t *>(O);
    a.sq_b = sq_b, a.sq_h = sq_h, a.sq_t = sq_t, a.sk_b = sk_b, a.sk_h = sk_h, a.sk_t = sk_t;
    a.sv_b = sv_b, a.sv_h = sv_h, a.sv_t = sv_t, a.so_b = so_b, a.so_h = so_h, a.so_t = so_t;
    a.H = int(heads), a.Tq = int(Tq), a.Tk = int(Tk), a.d = int(head_dim);
    a.causal = causal;
    // a head's queries on several workgroups once there are many (each stages K and V again): blocks of 32 queries, 4 waves
    const int64_t nblk = (Tq + 31) / 32;
    int64_t split = (nblk + 11) / 12;                                        // ~3 blocks per wave
    if (split > 65535) split = 65535;
    a.qsplit = int(split);
    // few queries per head (the T5 decoder's 16 tokens, the encoder's 64): 4 or 2 heads share a workgroup, each with its own
    // K / V image and its own waves, as long as the images fit
    {
        const int ds_ = int(head_dim <= 64 ? 2 : (head_dim + 31) / 32), kt_ = int(((Tk + 31) >> 5) << 1);
        const size_t per_head = size_t(2) * ((size_t(kt_) * 16 * (64 * ds_ + 16) + 1023) & ~size_t(1023));
        int hpw = nblk <= 1 ? 4 : (nblk <= 2 ? 2 : 1);
        while (hpw > 1 && per_head * hpw > size_t(72) * 1024) hpw >>= 1;      // (at least two workgroups per CU)
        a.hpw = hpw;
        a.nheads = int(batch * heads);
    }
    a.scale_log2e = scale * 1.44269504088896340736f;
    {
        static const bool dma_ok = [] {
            const char *e = getenv("VLMC_SDPA_DMA");                          // 0: K and V staged through registers (the cross-check)
            return !(e && e[0] == '0');
        }();
        const uintptr_t bits = reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) | uintptr_t(2 * (sk_b | sk_h | sk_t | sv_b | sv_h | sv_t));
        a.dma = dma_ok && (bits & 15u) == 0;
    }
    hipStream_t s = as_stream(stream);
    const int ds = int((head_dim + 31) / 32);
    const int rc = dtype == VLMC_F16 ? sdpa_dispatch_f16(a, batch * heads, ds, s) : sdpa_dispatch_bf16(a, batch * heads, ds, s);
    if (rc != VLMC_OK) return rc;
    VLMC_HIP_CHECK_LAUNCH("vlmc_sdpa_fwd");
    return VLMC_OK;
}
