// Compiled host path for the entry points a calibration replay calls thousands of times: vlmc_linear_fwd,
// vlmc_linear_fwd_group, vlmc_linear_fwd_rows, vlmc_attn_matmul, vlmc_row_mean, vlmc_sdpa_fwd and vlmc_rms_norm (include/vlmc.h).
//
// The replay of the reference's per-sample block forwards (wanda_pruner.py:308-311, :343-346) issues 1 000 - 15 000 of these
// launches per prune, most of them on a few hundred rows when the calibration text is ragged or the samples are sharded over
// GPUs: the kernels take 10-50 us and the ctypes route (vlmc/ops.py: argument checks, reshape, torch.empty, a job table,
// 12-20 converted arguments) 8-22 us of host time each -- the host sets the pace.  This module does the same checks, views and
// allocations through ATen and calls the SAME C ABI (it links libvlmc_hip.so; no kernel lives here): ~2 us per call.
// vlmc/ops.py uses it when it has been built (__graft_entry__.build() builds it) and falls back to ctypes otherwise -- the
// results are the library's either way.  PyTorch here is plumbing: tensor metadata, allocation, the autograd / autocast flags.
#include <torch/extension.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/vlmc.h"

namespace {

inline int dtype_code(at::ScalarType t) {
    if (t == at::kHalf) return VLMC_F16;
    if (t == at::kBFloat16) return VLMC_BF16;
    return -1;
}

inline void check(int rc) {
    if (rc != VLMC_OK) throw std::runtime_error(std::string("vlmc error ") + std::to_string(rc) + ": " + vlmc_last_error());
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// what vlmc/ops.py: linear_fwd_supported asks of (x, weight, bias)
bool linear_ok(const at::Tensor &x, const at::Tensor &w, const c10::optional<at::Tensor> &b) {
    if (!x.is_cuda() || !w.is_cuda() || x.scalar_type() != w.scalar_type() || dtype_code(x.scalar_type()) < 0) return false;
    if (w.dim() != 2 || x.dim() < 1 || x.size(-1) != w.size(1) || w.size(1) % 8 != 0 || w.stride(1) != 1 || w.stride(0) % 8 != 0 ||
        !aligned16(w.data_ptr()))
        return false;
    if (b.has_value() && b->defined()) {
        if (!b->is_cuda() || b->scalar_type() != w.scalar_type() || !b->is_contiguous()) return false;
    }
    return true;
}

// x as [M, K] rows the kernel can read: K-contiguous, 16-byte aligned rows
at::Tensor rows_of(const at::Tensor &x, int64_t K) {
    at::Tensor x2 = x.reshape({-1, K});
    if (x2.stride(1) != 1 || x2.stride(0) % 8 != 0 || x2.stride(0) < K || !aligned16(x2.data_ptr())) x2 = x2.contiguous();
    return x2;
}

std::vector<int64_t> lead_shape(const at::Tensor &x, int64_t N) {
    std::vector<int64_t> s(x.sizes().begin(), x.sizes().end() - 1);
    s.push_back(N);
    return s;
}

// fp32 (the reference's Q-Former: vlmc/ops.py: linear_f32_supported): the fp32 matrix-core kernel behind the same entry point
py::object linear_fwd_f32(const at::Tensor &x, const at::Tensor &w, const c10::optional<at::Tensor> &b, int64_t stream) {
    if (!x.is_cuda() || !w.is_cuda() || x.scalar_type() != at::kFloat || w.dim() != 2 || x.dim() < 1 || x.size(-1) != w.size(1) || w.size(1) == 0 ||
        w.stride(1) != 1)
        return py::none();
    const bool has_b = b.has_value() && b->defined();
    if (has_b && (!b->is_cuda() || b->scalar_type() != at::kFloat || !b->is_contiguous())) return py::none();
    const int64_t N = w.size(0), K = w.size(1);
    at::Tensor x2 = x.reshape({-1, K});
    if (x2.stride(1) != 1 || (x2.size(0) > 1 && x2.stride(0) < K)) x2 = x2.contiguous();
    const int64_t M = x2.size(0);
    at::Tensor y = at::empty({M, N}, x.options());
    check(vlmc_linear_fwd(x2.data_ptr(), w.data_ptr(), has_b ? b->data_ptr() : nullptr, VLMC_F32, M, N, K, M > 1 ? x2.stride(0) : K, w.stride(0),
                          y.data_ptr(), N, reinterpret_cast<void *>(stream)));
    return py::cast(y.reshape(lead_shape(x, N)));
}

py::object linear_fwd(const at::Tensor &x, const at::Tensor &w, const c10::optional<at::Tensor> &b, int64_t stream) {
    if (w.scalar_type() == at::kFloat) return linear_fwd_f32(x, w, b, stream);
    if (!linear_ok(x, w, b)) return py::none();
    const int64_t N = w.size(0), K = w.size(1);
    const at::Tensor x2 = rows_of(x, K);
    const int64_t M = x2.size(0);
    at::Tensor y = at::empty({M, N}, x.options());
    const void *bias = (b.has_value() && b->defined()) ? b->data_ptr() : nullptr;
    check(vlmc_linear_fwd(x2.data_ptr(), w.data_ptr(), bias, dtype_code(x.scalar_type()), M, N, K, x2.stride(0), w.stride(0), y.data_ptr(), N,
                          reinterpret_cast<void *>(stream)));
    return py::cast(y.reshape(lead_shape(x, N)));
}

py::object linear_fwd_group(const at::Tensor &x, const std::vector<at::Tensor> &ws, const std::vector<c10::optional<at::Tensor>> &bs,
                            int64_t stream) {
    const size_t n = ws.size();
    if (n < 1 || n > 4 || bs.size() != n) return py::none();
    for (size_t g = 0; g < n; ++g)
        if (!linear_ok(x, ws[g], bs[g]) || ws[g].size(1) != ws[0].size(1)) return py::none();
    const int64_t K = ws[0].size(1);
    const at::Tensor x2 = rows_of(x, K);
    const int64_t M = x2.size(0);
    vlmc_linear_job jobs[4];
    std::vector<at::Tensor> ys;
    ys.reserve(n);
    for (size_t g = 0; g < n; ++g) {
        const int64_t N = ws[g].size(0);
        ys.push_back(at::empty({M, N}, x.options()));
        jobs[g].W = ws[g].data_ptr();
        jobs[g].bias = (bs[g].has_value() && bs[g]->defined()) ? bs[g]->data_ptr() : nullptr;
        jobs[g].Y = ys.back().data_ptr();
        jobs[g].N = N;
        jobs[g].ldw = ws[g].stride(0);
        jobs[g].ldy = N;
    }
    check(vlmc_linear_fwd_group(x2.data_ptr(), jobs, int(n), dtype_code(x.scalar_type()), M, K, x2.stride(0), reinterpret_cast<void *>(stream)));
    py::list out;
    for (size_t g = 0; g < n; ++g) out.append(py::cast(ys[g].reshape(lead_shape(x, ws[g].size(0)))));
    return std::move(out);
}

// the same over a ROW MAP (a padded group of ragged calibration samples; vlmc/ops.py: linear_fwd_rows): 16-bit or fp32; None when the
// call is not one the kernel takes (the caller's ctypes route then raises with the reason)
py::object linear_fwd_rows(const at::Tensor &x, const std::vector<at::Tensor> &ws, const std::vector<c10::optional<at::Tensor>> &bs,
                           const at::Tensor &rowmap, int64_t n_real, int64_t stream) {
    const size_t n = ws.size();
    if (n < 1 || n > 4 || bs.size() != n || !x.is_cuda()) return py::none();
    const bool f32 = x.scalar_type() == at::kFloat;
    for (size_t g = 0; g < n; ++g) {
        if (ws[g].dim() != 2 || ws[g].size(1) != ws[0].size(1)) return py::none();
        if (f32) {
            const at::Tensor &w = ws[g];
            if (!w.is_cuda() || w.scalar_type() != at::kFloat || x.dim() < 1 || x.size(-1) != w.size(1) || w.size(1) == 0 || w.stride(1) != 1) return py::none();
            if (bs[g].has_value() && bs[g]->defined() && (!bs[g]->is_cuda() || bs[g]->scalar_type() != at::kFloat || !bs[g]->is_contiguous())) return py::none();
        } else if (!linear_ok(x, ws[g], bs[g])) {
            return py::none();
        }
    }
    const int64_t K = ws[0].size(1);
    at::Tensor x2 = x.reshape({-1, K});
    if (x2.stride(1) != 1 || x2.stride(0) < K || (!f32 && (x2.stride(0) % 8 != 0 || !aligned16(x2.data_ptr())))) x2 = x2.contiguous();
    const int64_t M = x2.size(0);
    if (!rowmap.is_cuda() || rowmap.scalar_type() != at::kInt || rowmap.dim() != 1 || rowmap.size(0) != M || !rowmap.is_contiguous() || n_real < 1 ||
        n_real > M)
        return py::none();
    vlmc_linear_job jobs[4];
    std::vector<at::Tensor> ys;
    ys.reserve(n);
    for (size_t g = 0; g < n; ++g) {
        const int64_t N = ws[g].size(0);
        ys.push_back(at::empty({M, N}, x.options()));
        jobs[g].W = ws[g].data_ptr();
        jobs[g].bias = (bs[g].has_value() && bs[g]->defined()) ? bs[g]->data_ptr() : nullptr;
        jobs[g].Y = ys.back().data_ptr();
        jobs[g].N = N;
        jobs[g].ldw = ws[g].stride(0);
        jobs[g].ldy = N;
    }
    check(vlmc_linear_fwd_rows(x2.data_ptr(), jobs, int(n), f32 ? VLMC_F32 : dtype_code(x.scalar_type()), M, K, x2.stride(0),
                               static_cast<const int32_t *>(rowmap.data_ptr()), n_real, reinterpret_cast<void *>(stream)));
    py::list out;
    for (size_t g = 0; g < n; ++g) out.append(py::cast(ys[g].reshape(lead_shape(x, ws[g].size(0)))));
    return std::move(out);
}

// torch.matmul(a, b) for the batched products of attention; None when vlmc_attn_matmul does not compute the call
// (the conditions of vlmc/ops.py: attn_matmul_plan)
py::object attn_matmul(const at::Tensor &a, const at::Tensor &b, int64_t stream) {
    const int64_t nd = a.dim();
    const bool f32 = a.scalar_type() == at::kFloat;                       // (the fp32 Q-Former's products: the fp32 kernel, <= 65535 matrices)
    if (nd != b.dim() || nd < 3 || nd > 4 || a.scalar_type() != b.scalar_type() || (!f32 && dtype_code(a.scalar_type()) < 0) || !a.is_cuda() ||
        !b.is_cuda())
        return py::none();
    const int64_t M = a.size(-2), K = a.size(-1), N = b.size(-1);
    if (K != b.size(-2) || M == 0 || N == 0 || K == 0) return py::none();
    if (a.stride(-1) != 1 && K != 1) return py::none();
    int64_t sbk, sbn;
    if (K == 1 || b.stride(-2) == 1) {
        sbk = 1;
        sbn = b.stride(-1);
    } else if (b.stride(-1) == 1 || N == 1) {
        sbk = b.stride(-2);
        sbn = 1;
    } else {
        return py::none();
    }
    if (a.stride(-2) < 0 || sbk < 0 || sbn < 0) return py::none();
    int64_t batch[2] = {1, 1}, sa[2] = {0, 0}, sb[2] = {0, 0};
    std::vector<int64_t> oshape;
    for (int64_t i = 0; i < nd - 2; ++i) {
        const int64_t x = a.size(i), y = b.size(i);
        if (x != y && x != 1 && y != 1) return py::none();
        const int64_t n = x != 1 ? x : y;
        if (n == 0) return py::none();
        const int64_t slot = i + (4 - nd);
        batch[slot] = n;
        sa[slot] = x != 1 ? a.stride(i) : 0;
        sb[slot] = y != 1 ? b.stride(i) : 0;
        oshape.push_back(n);
    }
    oshape.push_back(M);
    oshape.push_back(N);
    if (f32 && batch[0] * batch[1] > 65535) return py::none();
    at::Tensor out = at::empty(oshape, a.options());
    check(vlmc_attn_matmul(a.data_ptr(), b.data_ptr(), out.data_ptr(), f32 ? VLMC_F32 : dtype_code(a.scalar_type()), batch[0], batch[1], M, N, K, sa[0], sa[1],
                           a.stride(-2), sb[0], sb[1], sbk, sbn, batch[1] * M * N, M * N, N, reinterpret_cast<void *>(stream)));
    return py::cast(out);
}

// x.mean(-1, keepdim) of an fp32 CUDA tensor on vlmc_row_mean; None when the call is not one it takes
py::object row_mean(const at::Tensor &x, bool keepdim, int64_t stream) {
    if (!x.is_cuda() || x.scalar_type() != at::kFloat || x.dim() < 1 || x.size(-1) == 0) return py::none();
    const int64_t n = x.size(-1);
    at::Tensor x2 = x.reshape({-1, n});
    if (x2.stride(1) != 1 || (x2.size(0) > 1 && x2.stride(0) < n)) x2 = x2.contiguous();
    const int64_t rows = x2.size(0);
    std::vector<int64_t> oshape(x.sizes().begin(), x.sizes().end() - 1);
    if (keepdim) oshape.push_back(1);
    at::Tensor out = at::empty(oshape, x.options());
    check(vlmc_row_mean(static_cast<const float *>(x2.data_ptr()), rows, n, rows > 1 ? x2.stride(0) : n, static_cast<float *>(out.data_ptr()),
                        reinterpret_cast<void *>(stream)));
    return py::cast(out);
}

// F.scaled_dot_product_attention(q, k, v) on vlmc_sdpa_fwd; None when the kernel does not take the call (vlmc/ops.py: sdpa_plan)
py::object sdpa(const at::Tensor &q, const at::Tensor &k, const at::Tensor &v, double scale, bool causal, int64_t stream) {
    if (q.dim() != 4 || k.dim() != 4 || v.dim() != 4 || dtype_code(q.scalar_type()) < 0 || k.scalar_type() != q.scalar_type() ||
        v.scalar_type() != q.scalar_type() || !q.is_cuda() || !k.is_cuda() || !v.is_cuda())
        return py::none();
    const int64_t B = q.size(0), H = q.size(1), Tq = q.size(2), d = q.size(3), Tk = k.size(2);
    if (k.size(0) != B || k.size(1) != H || k.size(3) != d || v.size(0) != B || v.size(1) != H || v.size(2) != Tk || v.size(3) != d ||
        d % 8 != 0 || d > 128 || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || d <= 0)
        return py::none();
    if (q.stride(3) != 1 || k.stride(3) != 1 || v.stride(3) != 1 || q.stride(2) < 0 || k.stride(2) < 0 || v.stride(2) < 0) return py::none();
    if (Tk > vlmc_sdpa_max_keys(d) || !(scale > 0.0) || !(scale < 1e30)) return py::none();
    at::Tensor out = at::empty({B, Tq, H, d}, q.options());
    check(vlmc_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dtype_code(q.scalar_type()), B, H, Tq, Tk, d, q.stride(0), q.stride(1),
                        q.stride(2), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2), Tq * H * d, d, H * d, float(scale),
                        causal ? 1 : 0, reinterpret_cast<void *>(stream)));
    return py::cast(out.transpose(1, 2));
}

// weight * (x * rsqrt(mean(x.float()^2) + eps)).to(dtype) on vlmc_rms_norm; None when the call is not one it takes
py::object rms_norm(const at::Tensor &x, const at::Tensor &w, double eps, int64_t rsqrt_mode, int64_t stream) {
    if (!x.is_cuda() || !w.is_cuda() || dtype_code(x.scalar_type()) < 0 || w.scalar_type() != x.scalar_type() || x.dim() < 1 || w.dim() != 1 ||
        x.size(-1) != w.size(0) || w.size(0) == 0 || !w.is_contiguous())
        return py::none();
    const int64_t n = w.size(0);
    at::Tensor x2 = x.reshape({-1, n});
    if (x2.stride(1) != 1 || (x2.size(0) > 1 && x2.stride(0) < n)) x2 = x2.contiguous();
    const int64_t rows = x2.size(0);
    at::Tensor out = at::empty({rows, n}, x.options());
    check(vlmc_rms_norm(x2.data_ptr(), dtype_code(x.scalar_type()), rows, n, rows > 1 ? x2.stride(0) : n, w.data_ptr(), float(eps), int(rsqrt_mode),
                        out.data_ptr(), n, reinterpret_cast<void *>(stream)));
    return py::cast(out.reshape(x.sizes()));
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "compiled host path of vlmc_linear_fwd / vlmc_linear_fwd_group / vlmc_attn_matmul (same C ABI as the ctypes route)";
    m.def("linear_fwd", &linear_fwd, py::arg("x"), py::arg("weight"), py::arg("bias"), py::arg("stream"));
    m.def("linear_fwd_group", &linear_fwd_group, py::arg("x"), py::arg("weights"), py::arg("biases"), py::arg("stream"));
    m.def("attn_matmul", &attn_matmul, py::arg("a"), py::arg("b"), py::arg("stream"));
    m.def("row_mean", &row_mean, py::arg("x"), py::arg("keepdim"), py::arg("stream"));
    m.def("sdpa", &sdpa, py::arg("q"), py::arg("k"), py::arg("v"), py::arg("scale"), py::arg("causal"), py::arg("stream"));
    m.def("rms_norm", &rms_norm, py::arg("x"), py::arg("weight"), py::arg("eps"), py::arg("rsqrt_mode"), py::arg("stream"));
    m.def("linear_fwd_rows", &linear_fwd_rows);
    m.def("abi_version", []() { return vlmc_abi_version(); });
}
