// instantiations of csrc/attn_fused_kernel.hpp: bf16, 0 addend(s) (a translation unit of their own: `make -j` compiles them side by side)
#include "attn_fused_kernel.hpp"

namespace vlmc {
int attn_dispatch_bf16_0(const AttnArgs &a, int64_t bh, int ds, hipStream_t s) { return attn_dispatch<bf16_t, 0>(a, bh, ds, s); }
}  // namespace vlmc
