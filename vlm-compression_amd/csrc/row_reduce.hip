// vlmc_row_mean: batch-invariant mean over the last dimension of an fp32 matrix.
//
// The norms of the language models square their input in fp32 and average over the hidden dimension --
// `hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)` (transformers' T5LayerNorm, called from modeling_t5.py's
// blocks; LlamaRMSNorm likewise).  torch's reduction kernel picks its launch configuration -- how many threads and blocks share
// one output -- by the NUMBER of outputs: 4 rows (one calibration sample with a 4-token answer) are summed in another order
// than the same 4 rows inside a group of 512, and the last bit of the mean differs (measured: 72 of 512 rows of a Flan-T5-XL
// decoder block's first norm).  The reference replays every block one sample per forward (wanda_pruner.py:308-311, :343-346);
// the grouped replay must give a sample the bits its own forward would, so during a replay this reduction runs here: one
// wave per row, lane l adds elements 4 l + 256 i .. + 3 in ascending i, the 64 partial sums meet in a fixed butterfly, one
// IEEE division by n.  A row's mean depends on the row and n only.  HBM-bound (4 B read per element), one launch.
#include "common.hpp"

namespace vlmc {
namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

}  // namespace

__global__ __launch_bounds__(256) void row_mean_kernel(const float *__restrict__ x, int64_t rows, int n, int64_t ldx, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = x + row * ldx;
    float acc = 0.f;
    const bool vec = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    for (int c0 = 4 * lane; c0 < n; c0 += 256) {
        float v[4];
        if (vec && c0 + 3 < n) {
            const f32x4v q = *reinterpret_cast<const f32x4v *>(p + c0);
            v[0] = q[0], v[1] = q[1], v[2] = q[2], v[3] = q[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = c0 + j < n ? p[c0 + j] : 0.f;
        }
        acc = ieee_add(ieee_add(ieee_add(ieee_add(acc, v[0]), v[1]), v[2]), v[3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = ieee_add(acc, __shfl_xor(acc, off, kWave));
    if (lane == 0) out[row] = ieee_div(acc, float(n));
}

}  // namespace vlmc

using namespace vlmc;

extern "C" int vlmc_row_mean(const float *x, int64_t rows, int64_t n, int64_t ldx, float *out, void *stream) {
    VLMC_REQUIRE(x && out, "vlmc_row_mean: null pointer");
    VLMC_REQUIRE(rows >= 0 && n > 0 && n < (int64_t(1) << 30) && ldx >= n && rows < (int64_t(1) << 32), "vlmc_row_mean: bad shape");
    if (rows == 0) return VLMC_OK;
    hipLaunchKernelGGL(row_mean_kernel, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, rows, int(n), ldx, out);
    VLMC_HIP_CHECK_LAUNCH("vlmc_row_mean");
    return VLMC_OK;
}
