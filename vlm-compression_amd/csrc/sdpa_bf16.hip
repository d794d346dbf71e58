// bf16 instantiations of the fused attention kernel (csrc/sdpa_kernel.hpp; the C entry point is in sdpa.hip)
#include "sdpa_kernel.hpp"

namespace vlmc {
int sdpa_dispatch_bf16(const SdpaArgs &a, int64_t bh, int ds, hipStream_t s) { return sdpa_dispatch<bf16_t>(a, bh, ds, s); }
}  // namespace vlmc
