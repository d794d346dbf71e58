// vlmc_attn_fwd: the C entry point and the fp16 / no-addend instantiations of csrc/attn_fused_kernel.hpp (the kernel and its
// description live there; attn_fused_inst*.hip hold the other instantiations).
#include "attn_fused_kernel.hpp"

namespace vlmc {
int attn_dispatch_f16_0(const AttnArgs &a, int64_t bh, int ds, hipStream_t s) { return attn_dispatch<f16_t, 0>(a, bh, ds, s); }
}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_attn_max_keys(int64_t head_dim) {
    if (head_dim <= 0 || head_dim > 128 || head_dim % 8 != 0) return 0;
    return attn_max_keys(int(head_dim));
}

extern "C" int vlmc_attn_fwd(const void *Q, const void *K, const void *V, void *O, int dtype, int64_t batch, int64_t heads, int64_t Tq,
                             int64_t Tk, int64_t head_dim, const int64_t *q_strides, const int64_t *k_strides, const int64_t *v_strides,
                             int has_mul, float mul, const void *add0, const int64_t *add0_strides, const void *add1,
                             const int64_t *add1_strides, void *stream) {
    return vlmc_attn_fwd_lens(Q, K, V, O, dtype, batch, heads, Tq, Tk, head_dim, q_strides, k_strides, v_strides, has_mul, mul, add0,
                              add0_strides, add1, add1_strides, nullptr, nullptr, stream);
}

extern "C" int vlmc_attn_fwd_lens(const void *Q, const void *K, const void *V, void *O, int dtype, int64_t batch, int64_t heads, int64_t Tq,
                                  int64_t Tk, int64_t head_dim, const int64_t *q_strides, const int64_t *k_strides,
                                  const int64_t *v_strides, int has_mul, float mul, const void *add0, const int64_t *add0_strides,
                                  const void *add1, const int64_t *add1_strides, const int32_t *q_len, const int32_t *k_len, void *stream) {
    VLMC_REQUIRE(dtype == VLMC_F16 || dtype == VLMC_BF16, "vlmc_attn_fwd: dtype must be VLMC_F16 or VLMC_BF16");
    VLMC_REQUIRE(Q && K && V && O && q_strides && k_strides && v_strides, "vlmc_attn_fwd: null pointer");
    VLMC_REQUIRE(batch > 0 && heads > 0 && Tq > 0 && Tk > 0, "vlmc_attn_fwd: empty problem");
    VLMC_REQUIRE(head_dim > 0 && head_dim <= 128 && head_dim % 8 == 0, "vlmc_attn_fwd: head_dim must be a multiple of 8, at most 128 (got %lld)",
                 (long long)head_dim);
    VLMC_REQUIRE(Tk <= attn_max_keys(int(head_dim)), "vlmc_attn_fwd: %lld keys per head, at most %d for head_dim %lld (vlmc_attn_max_keys)",
                 (long long)Tk, attn_max_keys(int(head_dim)), (long long)head_dim);
    VLMC_REQUIRE(batch * heads < (int64_t(1) << 31) && Tq < (int64_t(1) << 24) && heads < (int64_t(1) << 20), "vlmc_attn_fwd: shape too large");
    VLMC_REQUIRE(has_mul == 0 || has_mul == 1, "vlmc_attn_fwd: has_mul must be 0 or 1");
    VLMC_REQUIRE(!has_mul || std::isfinite(mul), "vlmc_attn_fwd: the multiplier must be finite");
    VLMC_REQUIRE(add0 || !add1, "vlmc_attn_fwd: add1 without add0");
    VLMC_REQUIRE((!add0 || add0_strides) && (!add1 || add1_strides), "vlmc_attn_fwd: an addend needs its strides");
    for (int i = 0; i < 3; ++i)
        VLMC_REQUIRE(q_strides[i] >= 0 && k_strides[i] >= 0 && v_strides[i] >= 0, "vlmc_attn_fwd: negative stride");
    VLMC_REQUIRE(((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) |
                   reinterpret_cast<uintptr_t>(O) | reinterpret_cast<uintptr_t>(add0) | reinterpret_cast<uintptr_t>(add1)) & 1u) == 0,
                 "vlmc_attn_fwd: pointers must be 2-byte aligned");
    AttnArgs a{};
    a.Q = static_cast<const uint16_t *>(Q), a.K = static_cast<const uint16_t *>(K), a.V = static_cast<const uint16_t *>(V);
    a.O = static_cast<uint16_t *>(O);
    a.B0 = static_cast<const uint16_t *>(add0), a.B1 = static_cast<const uint16_t *>(add1);
    a.sq_b = q_strides[0], a.sq_h = q_strides[1], a.sq_t = q_strides[2];
    a.sk_b = k_strides[0], a.sk_h = k_strides[1], a.sk_t = k_strides[2];
    a.sv_b = v_strides[0], a.sv_h = v_strides[1], a.sv_t = v_strides[2];
    a.so_b = Tq * heads * head_dim, a.so_h = head_dim, a.so_t = heads * head_dim;              // O is [batch, Tq, heads, head_dim]
    if (add0) {
        for (int i = 0; i < 4; ++i) VLMC_REQUIRE(add0_strides[i] >= 0, "vlmc_attn_fwd: negative stride");
        VLMC_REQUIRE(add0_strides[3] >= 1 || Tk == 1, "vlmc_attn_fwd: an addend's key stride must be positive");
        a.s0_b = add0_strides[0], a.s0_h = add0_strides[1], a.s0_q = add0_strides[2], a.s0_k = add0_strides[3];
    }
    if (add1) {
        for (int i = 0; i < 4; ++i) VLMC_REQUIRE(add1_strides[i] >= 0, "vlmc_attn_fwd: negative stride");
        VLMC_REQUIRE(add1_strides[3] >= 1 || Tk == 1, "vlmc_attn_fwd: an addend's key stride must be positive");
        a.s1_b = add1_strides[0], a.s1_h = add1_strides[1], a.s1_q = add1_strides[2], a.s1_k = add1_strides[3];
    }
    a.H = int(heads), a.Tq = int(Tq), a.Tk = int(Tk), a.d = int(head_dim);
    a.has_mul = has_mul, a.mul = mul;
    VLMC_REQUIRE(k_len == nullptr || add0 != nullptr, "vlmc_attn_fwd_lens: k_len needs an addend that masks the keys behind it");
    a.qlen = q_len, a.klen = k_len;
    // a head's queries on several workgroups once there are many (each stages K and V again): blocks of 16 queries, ~6 per wave
    a.nblk = int((Tq + 15) / 16);
    a.bpw = 6;
    {
        static const bool dma_ok = [] {
            const char *e = getenv("VLMC_ATTN_DMA");                          // 0: K and V staged through registers (the cross-check)
            return !(e && e[0] == '0');
        }();
        const uintptr_t bits = reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) |
                               uintptr_t(2 * (a.sk_b | a.sk_h | a.sk_t | a.sv_b | a.sv_h | a.sv_t));
        a.dma = dma_ok && (bits & 15u) == 0;
    }
    hipStream_t s = as_stream(stream);
    const int ds = int((head_dim + 31) / 32);
    const int nadd = add1 ? 2 : add0 ? 1 : 0;
    const int64_t bh = batch * heads;
    const int rc = dtype == VLMC_F16 ? (nadd == 0 ? attn_dispatch_f16_0(a, bh, ds, s) : nadd == 1 ? attn_dispatch_f16_1(a, bh, ds, s) : attn_dispatch_f16_2(a, bh, ds, s))
                                     : (nadd == 0 ? attn_dispatch_bf16_0(a, bh, ds, s) : nadd == 1 ? attn_dispatch_bf16_1(a, bh, ds, s) : attn_dispatch_bf16_2(a, bh, ds, s));
    if (rc != VLMC_OK) return rc;
    VLMC_HIP_CHECK_LAUNCH("vlmc_attn_fwd");
    return VLMC_OK;
}
