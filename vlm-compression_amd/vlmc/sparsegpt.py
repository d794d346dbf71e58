"""SparseGPT on the GPU: Hessian accumulation, inverse factor, blocked OBS pruning.

Mirrors the `SparseGPT` helper class of lavis/compression/pruners/sparsegpt_pruner.py:53-219
(`add_batch`, `fasterprune`, `free`).  Division of labour:

* Hessian `H = (2/n) X^T X` running mean (:68-79): one library GEMM per hook call
  (`addmm_` with beta = n/(n+b) -- scale and accumulate fused);
* dead columns, inf clamps, damp-only-on-failure Cholesky, `cholesky_inverse`, upper Cholesky
  (:92-160): the library factorizations (rocSOLVER through torch), `cholesky_ex` instead of
  exception handling so that nothing synchronises except the failure check itself;
* per 128-column block (:167-210): unstructured mask by the block threshold (elementwise +
  library sort, bit-identical scores), then the sequential column sweep in ONE fused kernel
  (`vlmc_sparsegpt_sweep`, csrc/sparsegpt.hip) instead of ~1300 tiny launches, then the trailing
  update as a library GEMM.

Parity bar (BASELINE.json): masks identical up to near-ties, weights within 1e-3 relative of the
reference's fp32 result (the factorization and the GEMMs accumulate in a different order).
"""
from __future__ import annotations

import math

import torch

from . import _lib
from . import ops
from .ops import _need_gpu, _stream


_DEFER = __import__("os").environ.get("VLMC_SGPT_DEFER", "1") != "0"
_SYRK = __import__("os").environ.get("VLMC_SGPT_SYRK", "1") != "0"
# fp32 activations (toy / full-precision models) go as nine bf16 plane products: exact products, but 9x the matrix-core work
# and a scattering transpose -- measured 48 TFLOP/s-equivalent against 129 for the library's fp32 GEMM, so off by default
_SYRK_F32 = __import__("os").environ.get("VLMC_SGPT_SYRK_F32", "0") == "1"
_DEFER_BYTES = 256 << 20        # staged activations (as fp32) folded into H once they exceed this


class SparseGPT:
    """Same surface as the reference helper: `add_batch(inp, out)`, `fasterprune(...)`, `free()`.

    The reference's recurrence (:68-79) treats a hook input of batch b as ONE update
    `H <- H * n/(n+b) + (2/(n+b)) X^T X` over all its tokens.  With batch-1 calibration samples that is 128 tiny
    GEMMs (+ a cast and a scale) per distinct input and block -- 54k host-bound launch triples over the model.
    Here consecutive calls are staged and folded in with the same formula as if they had arrived as one larger batch
    (`VLMC_SGPT_DEFER=0`: one update per call, the reference's exact sequence); `H` folds what is pending."""

    def __init__(self, layer):
        self.layer = layer
        self.dev = layer.weight.device
        _need_gpu(layer.weight)
        self.rows, self.columns = layer.weight.shape
        self._H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0
        self._folded = 0                     # samples already inside _H
        self._staged, self._staged_elems = [], 0
        self._lower_only = False
        self.factor_cache = {}

    @property
    def H(self):
        self._fold()
        if self._lower_only:                  # the SYRK kernel maintains the tiles on and below the diagonal
            ops.symmetrize_lower(self._H)
            self._lower_only = False
        return self._H

    @H.setter
    def H(self, value):
        self._staged, self._staged_elems = [], 0
        self._lower_only = False
        self._H = value

    def _accumulate(self, X, alpha, beta):
        """H <- alpha H + beta X^T X (:76-79).  `vlmc_hessian_accum` (hand-written MFMA SYRK: 16-bit products are exact in
        fp32; lower-triangle tiles only) for 16-bit activations; fp32 activations and `VLMC_SGPT_SYRK=0`: the library GEMM."""
        if _SYRK and (X.dtype != torch.float32 or _SYRK_F32):
            ops.hessian_accum(self._H, X, alpha, beta)
            self._lower_only = True
        else:
            Xf = X.float()
            self._H.addmm_(Xf.t(), Xf, beta=alpha, alpha=beta)

    @torch.no_grad()
    def _fold(self):
        if not self._staged:
            return
        n = self.nsamples
        X = torch.cat(self._staged, dim=0) if len(self._staged) > 1 else self._staged[0]
        self._accumulate(X, self._folded / n, 2.0 / n)                       # :76-79 with b = the staged samples
        self._folded = n
        self._staged, self._staged_elems = [], 0

    @torch.no_grad()
    def add_batch(self, inp, out=None):
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        x = inp.reshape(-1, inp.shape[-1])
        # rows of X seen so far: X^T X of fewer rows than columns is singular, which decides the factorization route
        self.factor_cache["rows_seen"] = self.factor_cache.get("rows_seen", 0) + x.shape[0]
        if _DEFER:
            self.nsamples += b
            if not self._staged and x.numel() * 4 > _DEFER_BYTES:
                # a grouped forward hands over all its samples in ONE call: folded in at once with the very formula `_fold`
                # would apply to it alone -- no copy of the activations (92-404 MB per ViT-g input) is kept
                n = self.nsamples
                self._accumulate(x, self._folded / n, 2.0 / n)                  # :76-79 with b = this call's samples
                self._folded = n
                return
            self._staged.append(x.clone())               # the caller may reuse the activation's memory (graph replay)
            self._staged_elems += x.numel()
            if self._staged_elems * 4 > _DEFER_BYTES:
                self._fold()
            return
        beta = self.nsamples / (self.nsamples + b)
        self.nsamples += b
        self._folded = self.nsamples
        if _SYRK and (x.dtype != torch.float32 or _SYRK_F32):
            self._accumulate(x, beta, 2.0 / self.nsamples)            # the scale 2/n rides in the epilogue
        else:
            xs = math.sqrt(2 / self.nsamples) * x.float()             # :78 (scaled in fp32 before the product)
            self._H.addmm_(xs.t(), xs, beta=beta, alpha=1.0)          # H *= beta; H += xs^T xs  (:76-79)

    def free(self):
        self.H = None
        self.factor_cache = {}


def _clamp_inf(H):
    pos = torch.isinf(H) & (H > 0)
    if bool(pos.any()):
        H[pos] = torch.quantile(H.flatten()[: 1 << 24], 0.999)
    neg = torch.isinf(H) & (H < 0)
    if bool(neg.any()):
        H[neg] = torch.quantile(H.flatten()[: 1 << 24], 0.001)


_CHOL_NB = 128
_CHOL_PANEL_GEMM = __import__("os").environ.get("VLMC_CHOL_PANEL_GEMM", "1") == "1"   # 0: library triangular solve for the panel
_CHOL_GRAPH = __import__("os").environ.get("VLMC_CHOL_GRAPH", "1") == "1"


def _chol_steps(A, L, inv, info):
    """The right-looking sweep over 128-column blocks on row-major fp32 buffers (A is consumed).
    `inv`: [128, 128] scratch, or [blocks, 128, 128] to keep the inverse of every diagonal block."""
    n = A.shape[0]
    lib = _lib.load()
    el = A.element_size()
    for k in range(0, n, _CHOL_NB):
        nb = min(_CHOL_NB, n - k)
        off = (k * n + k) * el
        inv_k = inv[k // _CHOL_NB] if inv.dim() == 3 else inv
        _lib.check(lib.vlmc_chol_block(A.data_ptr() + off, n, nb, L.data_ptr() + off, n, inv_k.data_ptr(), _CHOL_NB,
                                       info.data_ptr(), k, _stream()))
        if k + nb < n:
            L21 = L[k + nb:, k:k + nb]
            if _CHOL_PANEL_GEMM:
                torch.mm(A[k + nb:, k:k + nb], inv_k[:nb, :nb].t(), out=L21)            # = A21 inv(L11)^T, straight into L
            else:
                L21.copy_(torch.linalg.solve_triangular(L[k:k + nb, k:k + nb], A[k + nb:, k:k + nb].t(), upper=False).t())
            A[k + nb:, k + nb:].addmm_(L21, L21.t(), beta=1.0, alpha=-1.0)   # trailing update (the lower part is what is read)


_chol_graphs = {}       # (n, device index, slot) -> (graph, A, L, inv, info): the sweep is launch-bound when issued from Python


def _chol_graph(n, dev, slot):
    """The captured sweep for n x n matrices with its work buffers; `slot`: an instance of its own for every chain that may
    run at the same time as another of the same size (factorize_many).  Captured on first use -- from the calling thread."""
    key = (n, dev.index, slot)
    ent = _chol_graphs.get(key)
    if ent is None:
        A = torch.empty((n, n), dtype=torch.float32, device=dev)
        L = torch.zeros((n, n), dtype=torch.float32, device=dev)
        inv = torch.empty((_CHOL_NB, _CHOL_NB), dtype=torch.float32, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        A.copy_(torch.eye(n, device=dev))
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                      # one eager run before capture (library workspaces)
            _chol_steps(A, L, inv, info)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            _chol_steps(A, L, inv, info)
        cur.wait_stream(torch.cuda.current_stream(dev))
        ent = _chol_graphs[key] = (graph, A, L, inv, info)
    return ent


@torch.no_grad()
def blocked_cholesky(H: torch.Tensor, upper=False, slot: int = 0):
    """(factor, info) like torch.linalg.cholesky_ex(H, upper=upper) for a symmetric fp32 matrix on the GPU.
    Right-looking, 128-column blocks: the diagonal block and its inverse in ONE one-workgroup kernel
    (`vlmc_chol_block`), the panel below it and the trailing update as library GEMMs; the ~5 launches per block
    are captured once per matrix size in a HIP graph (`VLMC_CHOL_GRAPH=0` issues them eagerly)."""
    _need_gpu(H)
    assert H.dim() == 2 and H.shape[0] == H.shape[1] and H.dtype == torch.float32
    n = H.shape[0]
    dev = H.device
    if _CHOL_GRAPH and n > _CHOL_NB:
        graph, A, L, inv, info = _chol_graph(n, dev, slot)
        A.copy_(H)                                              # also converts a column-major H
        info.zero_()
        graph.replay()
        return (L.t().contiguous() if upper else L.clone()), info.clone()
    # row-major working copy (the library hands back column-major results, e.g. cholesky_inverse); the caller's H
    # must survive a failed attempt
    A = H.clone(memory_format=torch.contiguous_format)
    L = torch.zeros((n, n), dtype=torch.float32, device=dev)
    inv = torch.empty((_CHOL_NB, _CHOL_NB), dtype=torch.float32, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    _chol_steps(A, L, inv, info)
    return (L.t().contiguous() if upper else L), info


def release_caches():
    """Drop the captured factorization graphs and their n x n work buffers (4 + 2 fp32 matrices per distinct n: ~2 GB
    after a Vicuna-7B prune).  The SparseGPT pruners call it when `prune()` ends, so a RESSA training or evaluation stage in
    the same process starts without them; the next prune captures again."""
    _chol_graphs.clear()
    _inv_graphs.clear()
    _persist_bufs.clear()


_SELECT_THRESHOLD = __import__("os").environ.get("VLMC_SGPT_SORT_THRESHOLD", "0") != "1"
factor_stats = {"direct": 0, "chain": 0, "damped": 0}     # how often each route produced the inverse factor (damped: k >= 1 failed attempts, one factorization of H + k damp I)
_DIRECT_FACTOR = __import__("os").environ.get("VLMC_SGPT_DIRECT_FACTOR", "1") == "1"
_inv_graphs = {}        # (n, device index, slot) -> (graph, A, L, inv, X, U, info, fork stream)
# the inverse's rows on a second stream inside the graph (off: measured 3.9 s against 3.2 s per prune -- the cross-stream edges of a
# captured graph cost more than the two products they take off the critical path)
_INVERSE_FORK = __import__("os").environ.get("VLMC_SGPT_INVERSE_FORK", "0") == "1"


def _inverse_factor_steps(A, L, inv, X, U, info, side=None):
    """A = H with rows and columns reversed (consumed).  M = chol(A) (lower), X = M^-1 by block rows
    (X[i, :i] = -inv(M_ii) (M[i, :i] X[:i, :i]), the diagonal-block inverses come from vlmc_chol_block), and
    U = X with rows and columns reversed: upper triangular with U^T U = H^-1."""
    # One pass: the diagonal-block kernel writes inv(M_kk) straight into X's diagonal block, the panel product goes straight
    # into L, and block row k of X follows as soon as block row k of M is final (it is: right-looking) -- five graph nodes per
    # 128 columns (the separate factor / inverse passes with their slice copies were ten; a node costs 5-10 us of dispatch
    # on top of its kernel, and the chain is nothing but dependent nodes).
    # `side`: a second stream for the inverse's block rows.  Row k of the inverse needs inv(M_kk), row k of M (final once
    # the panels of the earlier steps are written) and the rows of the inverse above it -- nothing of the trailing update,
    # which is the factorization's critical path: the two products of row k run beside the panel and trailing GEMMs of
    # steps k, k + 1, .. (forked after the diagonal-block kernel, joined at the end; captured into the graph as such).
    n = A.shape[0]
    lib = _lib.load()
    el = A.element_size()
    X.zero_()
    main = torch.cuda.current_stream(A.device)
    if side is not None:
        side.wait_stream(main)
    for k in range(0, n, _CHOL_NB):
        nb = min(_CHOL_NB, n - k)
        off = (k * n + k) * el
        ik = X[k:k + nb, k:k + nb]
        _lib.check(lib.vlmc_chol_block(A.data_ptr() + off, n, nb, L.data_ptr() + off, n, X.data_ptr() + off, n, info.data_ptr(), k,
                                       _stream()))
        if k + nb < n:
            L21 = L[k + nb:, k:k + nb]
            torch.mm(A[k + nb:, k:k + nb], ik.t(), out=L21)                 # = A21 inv(M_kk)^T
        if k:
            row = X[k:k + nb, :k]
            if side is not None:
                side.wait_stream(main)                                      # inv(M_kk) and (from step k - 1's panel) row k of M
                with torch.cuda.stream(side):
                    torch.addmm(row, ik, torch.mm(L[k:k + nb, :k], X[:k, :k]), beta=0.0, alpha=-1.0, out=row)
            else:
                torch.addmm(row, ik, torch.mm(L[k:k + nb, :k], X[:k, :k]), beta=0.0, alpha=-1.0, out=row)
        if k + nb < n:
            A[k + nb:, k + nb:].addmm_(L21, L21.t(), beta=1.0, alpha=-1.0)
    if side is not None:
        main.wait_stream(side)
    U.copy_(torch.flip(X, (0, 1)))


_PERSISTENT = __import__("os").environ.get("VLMC_SGPT_PERSISTENT", "1") != "0"
_persist_bufs = {}      # (n, device index, slot) -> (A, M, X, U, info, workspace)
PERSISTENT_MAX_WORKGROUPS = int(__import__("os").environ.get("VLMC_SGPT_PERSISTENT_WGS", "128"))


def persistent_factor_usable(n: int) -> bool:
    """`vlmc_chol_inverse` takes n a multiple of 128 (every width of ViT-g / Flan-T5-XL / Vicuna-7B); `VLMC_SGPT_PERSISTENT=0`
    keeps the chain of launches per 128 columns (the cross-check)."""
    return _PERSISTENT and n % _CHOL_NB == 0 and n >= 2 * _CHOL_NB


def _inverse_upper_factor_persistent(H: torch.Tensor, slot: int, max_workgroups: int | None = None):
    """(U, info) from ONE persistent launch (csrc/chol_persistent.hip): A = J H J, M M^T = A, X = M^-1, U = J X J."""
    n, dev = H.shape[0], H.device
    key = (n, dev.index, slot)
    ent = _persist_bufs.get(key)
    lib = _lib.load()
    if ent is None:
        A = torch.empty((n, n), dtype=torch.float32, device=dev)
        M = torch.empty((n, n), dtype=torch.float32, device=dev)
        X = torch.zeros((n, n), dtype=torch.float32, device=dev)       # tiles above the diagonal are never written: stay zero
        U = torch.empty((n, n), dtype=torch.float32, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(int(lib.vlmc_chol_inverse_workspace(n)), dtype=torch.uint8, device=dev)
        ent = _persist_bufs[key] = (A, M, X, U, info, ws)
    A, M, X, U, info, ws = ent
    A.copy_(torch.flip(H, (0, 1)))
    info.zero_()
    wgs = PERSISTENT_MAX_WORKGROUPS if max_workgroups is None else int(max_workgroups)
    _lib.check(lib.vlmc_chol_inverse(A.data_ptr(), n, n, M.data_ptr(), n, X.data_ptr(), n, info.data_ptr(), ws.data_ptr(), ws.numel(),
                                     wgs, _stream()))
    U.copy_(torch.flip(X, (0, 1)))
    return U.clone(), info.clone()


@torch.no_grad()
def inverse_upper_factor(H: torch.Tensor, slot: int = 0, max_workgroups: int | None = None):
    """(U, info): the upper triangular U with U^T U = H^-1, i.e. what `cholesky(cholesky_inverse(cholesky(H)), upper=True)`
    (sparsegpt_pruner.py:112-150) arrives at, from ONE factorization: with J the index reversal, J H J = M M^T gives
    H = R R^T for the upper triangular R = J M J, hence H^-1 = (R^-1)^T R^-1 and U = R^-1 = J M^-1 J (the Cholesky
    factor is unique).  n^3/3 + n^3/3 flops instead of 4 n^3/3, one sequential sweep of diagonal blocks instead of two.
    info != 0: H is not (numerically) positive definite."""
    _need_gpu(H)
    assert H.dim() == 2 and H.shape[0] == H.shape[1] and H.dtype == torch.float32
    n = H.shape[0]
    dev = H.device
    if persistent_factor_usable(n):
        return _inverse_upper_factor_persistent(H, slot, max_workgroups)
    key = (n, dev.index, slot)           # `slot`: chains of equal size that run at the same time need buffers of their own
    ent = _inv_graphs.get(key)
    if ent is None:
        nblk = (n + _CHOL_NB - 1) // _CHOL_NB
        A = torch.empty((n, n), dtype=torch.float32, device=dev)
        L = torch.zeros((n, n), dtype=torch.float32, device=dev)
        X = torch.zeros((n, n), dtype=torch.float32, device=dev)
        U = torch.zeros((n, n), dtype=torch.float32, device=dev)
        inv = torch.zeros((nblk, _CHOL_NB, _CHOL_NB), dtype=torch.float32, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        graph = None
        fork = torch.cuda.Stream(device=dev) if _INVERSE_FORK else None
        if _CHOL_GRAPH and n > _CHOL_NB:
            A.copy_(torch.flip(H, (0, 1)))
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                      # one eager run before capture (library workspaces)
                _inverse_factor_steps(A, L, inv, X, U, info, fork)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                _inverse_factor_steps(A, L, inv, X, U, info, fork)
        ent = _inv_graphs[key] = (graph, A, L, inv, X, U, info, fork)
    graph, A, L, inv, X, U, info, fork = ent
    A.copy_(torch.flip(H, (0, 1)))
    info.zero_()
    if graph is not None:
        graph.replay()
    else:
        _inverse_factor_steps(A, L, inv, X, U, info, fork)
    return U.clone(), info.clone()


def _chol_with_damping(H, damp, upper, max_tries=100, slot=0):
    for _ in range(max_tries):
        L, info = blocked_cholesky(H, upper=upper, slot=slot)
        if int(info.item()) == 0 and not bool(torch.isnan(L).any()):
            return L
        H.diagonal().add_(damp)                                        # only after a failure (:114-128)
    raise _lib.VlmcError(_lib.VLMC_ENOTPD, "Hessian not positive definite after %d damping steps" % max_tries)


_FACTOR_STREAMS = {}       # device index -> side streams of factorize_many
_SWEEP_STREAMS = {}        # device index -> side streams for the column sweeps of a block's independent linears


def sweep_streams(dev):
    """Side streams for the sweeps of a block's linears (`VLMC_SGPT_SWEEP_STREAMS=n`, default 4; 1: none)."""
    n = int(__import__("os").environ.get("VLMC_SGPT_SWEEP_STREAMS", "4"))
    if n <= 1:
        return []
    sts = _SWEEP_STREAMS.setdefault(dev.index, [])
    while len(sts) < n:
        sts.append(torch.cuda.Stream(device=dev))
    return sts[:n]


def concurrent_factor_enabled():
    """`VLMC_SGPT_CONCURRENT=0`: the Hessians of a block are factorized one after the other, each with its own host checks."""
    return __import__("os").environ.get("VLMC_SGPT_CONCURRENT", "1") != "0"


@torch.no_grad()
def factorize_many(items, percdamp=0.01, max_streams=8, history=None):
    """Factorize the Hessians of one transformer block TOGETHER.  `items`: [(H, factor_cache)], one per distinct linear
    input (4-7 per Flan-T5-XL / ViT-g block); fills `factor_cache["U"]`, `["dead"]`, where `fasterprune` finds them.

    Each factorization is a serial chain of one-workgroup diagonal-block kernels and small GEMMs that leaves most of the
    chip idle, and the chains of a block are independent, so every chain runs on a side stream of its own (chains of equal
    size in separate buffer slots) and the host reads the outcome of ALL of them -- LAPACK info, NaN in the factor, +-inf in
    H -- in ONE copy (the one-by-one route makes five synchronising checks per Hessian).

    Damping (sparsegpt_pruner.py:112-150: `cholesky(H)`, retried with `H += damp I` while it fails, `cholesky_inverse`,
    `cholesky(., upper)`).  With k the number of failed attempts, the reference's result is the upper Cholesky factor of
    (H + k damp I)^-1 -- which `inverse_upper_factor` gives from ONE factorization (the factor is unique).  The attempts
    k = 0, 1, .. are therefore launched side by side as independent chains (damp = percdamp mean(diag H) is formed on the
    device; k = 0, 1 first, the next two in a second round if neither is clean) and the first clean one in k order is
    taken: what the reference's loop would have stopped at, without a host decision between attempts.  `history` (a dict the caller keeps per tower) remembers which inputs needed damping in
    the previous block: an input that did not is tried undamped only (the T5 decoder's 16-token samples give Hessians of
    fewer rows than columns in every block, a ViT block's never do).  An input whose attempts all fail -- and one whose
    attempt factorizes without a failing pivot but leaves NaN in the inverse factor (the reference's first loop passes such
    an H and its second loop damps Hinv) -- takes `factorize`, the reference's three steps with both loops, unchanged; the
    dead-column mask is the one computed here, before the diagonal was filled.
    Hessians of no more rows than columns (`rows_seen <= n`: singular, or nearly) stay on THIS route, unlike in `factorize`:
    a singular H fails at k = 0 and is clean at k = 1 -- the reference's own decision on its rank-deficient golden
    (tests/test_oracle_golden.py::test_sparsegpt_damping_route_of_the_reference_on_the_rank_deficient_golden: one damping step
    on H, none on Hinv) and fp64's; tests/test_sparsegpt_gpu.py holds the route taken here against both."""
    todo = [(H, c) for H, c in items if "U" not in c]
    if not todo:
        return
    if not (concurrent_factor_enabled() and _DIRECT_FACTOR and todo[0][0].is_cuda):
        for H, c in todo:
            c["U"], c["dead"] = factorize(H, percdamp, rows_seen=c.get("rows_seen"))
        return
    dev = todo[0][0].device
    main = torch.cuda.current_stream(dev)
    streams = _FACTOR_STREAMS.setdefault(dev.index, [])
    while len(streams) < max_streams:
        streams.append(torch.cuda.Stream(device=dev))
    history = {} if history is None else history
    slots, n_launched = {}, [0]

    budget = {}             # n -> workgroups a persistent factorization of that size may take (they share the chip's CUs)

    def plan_budget(sizes):
        """One persistent workgroup holds a whole CU (132 KB of LDS): the chains launched together share the CUs in proportion
        to their tile counts (n / 128)^2, at least 8 each -- the largest chain is throughput-bound below ~128-256 workgroups,
        the small ones are bound by their critical path whatever they get (tools/chain_probe2.py)."""
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        tot = sum((n // _CHOL_NB) ** 2 for n in sizes) or 1
        budget.clear()
        for n in set(sizes):
            tiles = (n // _CHOL_NB) ** 2
            budget[n] = max(8, min(tiles, (cus * tiles) // tot))

    def launch(H, k, damp):
        n = H.shape[0]
        slot = slots.get(n, 0)
        slots[n] = slot + 1
        st = streams[n_launched[0] % len(streams)]
        n_launched[0] += 1
        st.wait_stream(main)
        with torch.cuda.stream(st):
            if k:
                Hk = H.clone()
                Hk.diagonal().add_(damp * float(k))             # k failed attempts: H += damp, k times (:118-121)
            else:
                Hk = H
            U, info = inverse_upper_factor(Hk, slot=slot, max_workgroups=budget.get(n))
            failed = (info != 0).any()
            nan = torch.isnan(U).any()
            # [not clean, "factorized without a failing pivot but the inverse factor holds NaN"]: the second is not a case
            # for more damping of H -- the reference's first loop would pass such an H and damp Hinv in its second (:139-150)
            bad = torch.stack([failed | nan | torch.isinf(Hk).any(), nan & ~failed])
        for t in (U, bad):
            t.record_stream(main)
        return U, bad

    prepared = []
    for idx, (H, c) in enumerate(todo):
        dead = torch.diag(H) == 0
        H.diagonal().masked_fill_(dead, 1.0)                     # H[dead, dead] = 1 (:99-100) without the host round trip
        damp = percdamp * torch.mean(torch.diag(H))              # (:110) a device scalar
        ks = (0,) if history.get(idx, 1) == 0 else (0, 1)
        prepared.append((H, c, dead, damp, ks))
    plan_budget([H.shape[0] for H, c, dead, damp, ks in prepared for _ in ks])
    # (the largest chains first: a persistent grid that finds no free CU waits for one)
    order = sorted(range(len(prepared)), key=lambda i: -prepared[i][0].shape[0])
    attempts = [None] * len(prepared)
    for i in order:
        H, c, dead, damp, ks = prepared[i]
        attempts[i] = [(k,) + launch(H, k, damp) for k in ks]
    for st in streams:
        main.wait_stream(st)
    flags = torch.stack([bad for att in attempts for _, _, bad in att]).cpu().tolist()       # the ONE host read
    pos, retry = 0, []

    def reference_chain(idx, H, c, dead):
        # the reference's three steps with both damping loops, from the undamped H (its dead diagonal already filled: the
        # mask computed up front is the one to keep -- `factorize` would find none)
        history[idx] = 3
        U, _ = factorize(H, percdamp, rows_seen=c.get("rows_seen"), try_direct=False)
        c["U"], c["dead"] = U, dead

    for idx, ((H, c, dead, damp, ks), att) in enumerate(zip(prepared, attempts)):
        fl = flags[pos:pos + len(att)]
        pos += len(att)
        first_ok = next((i for i, f in enumerate(fl) if not f[0]), None)
        if any(f[1] for f in (fl if first_ok is None else fl[:first_ok])):
            reference_chain(idx, H, c, dead)                     # an earlier attempt factorized but its inverse holds NaN
        elif first_ok is not None:
            k = att[first_ok][0]
            factor_stats["direct" if k == 0 else "damped"] = factor_stats.get("direct" if k == 0 else "damped", 0) + 1
            c["U"], c["dead"] = att[first_ok][1], dead
            history[idx] = k
        else:
            retry.append((idx, H, c, dead, damp, ks))
    if retry:                                                    # the next two attempts of what has not come out clean
        plan_budget([H.shape[0] for idx, H, c, dead, damp, ks in retry for _ in range(2)])
        second = [(idx, H, c, dead, [(k,) + launch(H, k, damp) for k in (ks[-1] + 1, ks[-1] + 2)]) for idx, H, c, dead, damp, ks in retry]
        for st in streams:
            main.wait_stream(st)
        flat = [bad for *_, att in second for _, _, bad in att]
        fl2 = torch.stack(flat).cpu().tolist() if flat else []
        pos = 0
        for idx, H, c, dead, att in second:
            fl = fl2[pos:pos + len(att)]
            pos += len(att)
            first_ok = next((i for i, f in enumerate(fl) if not f[0]), None)
            if first_ok is not None and not any(f[1] for f in fl[:first_ok]):
                factor_stats["damped"] = factor_stats.get("damped", 0) + 1
                c["U"], c["dead"] = att[first_ok][1], dead
                history[idx] = att[first_ok][0]
            else:                                                # more than three damping steps, or a factor with NaN: the reference's route
                reference_chain(idx, H, c, dead)


@torch.no_grad()
def factorize(H: torch.Tensor, percdamp=0.01, rows_seen=None, slot=0, try_direct=True):
    """(U, dead): upper Cholesky factor of H^-1 (`Hinv`, :92-160) and the dead-column mask; consumes H.
    Depends on H only, so linears fed by the same tensor (q/k/v, wi_0/wi_1) share one factorization."""
    dead = torch.diag(H) == 0
    H.diagonal().masked_fill_(dead, 1.0)          # H[dead, dead] = 1 (:99-100): the diagonal entries, no host round trip
    _clamp_inf(H)
    if try_direct and _DIRECT_FACTOR and (rows_seen is None or rows_seen > H.shape[0]):
        # (a Hessian of no more rows than columns is singular or nearly so: straight to the reference's chain)
        # one factorization of the index-reversed Hessian instead of cholesky -> cholesky_inverse -> cholesky: the same
        # matrix (the factor is unique) with other roundings.  A Hessian that is not positive definite takes the
        # reference's three-step chain below with its two damping loops, unchanged.
        U, info = inverse_upper_factor(H, slot=slot)
        if int(info.item()) == 0 and not bool(torch.isnan(U).any()):
            factor_stats["direct"] += 1
            return U, dead
    factor_stats["chain"] += 1
    L = _chol_with_damping(H, percdamp * torch.mean(torch.diag(H)), upper=False, slot=slot)
    Hi = torch.cholesky_inverse(L)
    _clamp_inf(Hi)
    U = _chol_with_damping(Hi, percdamp * torch.mean(torch.diag(Hi).abs()), upper=True, slot=slot)
    return U.contiguous(), dead                          # the solver hands back a column-major factor


@torch.no_grad()
def inverse_factor(H: torch.Tensor, W: torch.Tensor, percdamp=0.01) -> torch.Tensor:
    """Upper Cholesky factor of H^-1; consumes H, zeroes W's dead columns (:92-160)."""
    U, dead = factorize(H, percdamp)
    W.masked_fill_(dead.unsqueeze(0), 0.0)
    return U


def sweep_block(W: torch.Tensor, i1: int, i2: int, U: torch.Tensor, mask1, prune_n, prune_m, err: torch.Tensor,
                mask_out: torch.Tensor | None = None):
    """In-place column sweep of W[:, i1:i2] (fp32) with U[i1:i2, i1:i2]; fills err[:, :i2-i1]."""
    _need_gpu(W, U, err, mask1, mask_out)
    assert W.dtype == torch.float32 and U.dtype == torch.float32 and err.dtype == torch.float32
    assert W.stride(1) == 1 and U.stride(1) == 1 and err.stride(1) == 1
    count = i2 - i1
    el = W.element_size()
    _lib.check(_lib.load().vlmc_sparsegpt_sweep(
        W.data_ptr() + i1 * el, W.shape[0], count, W.stride(0), U.data_ptr() + (i1 * U.stride(0) + i1) * el, U.stride(0),
        mask1.data_ptr() if mask1 is not None else None, mask1.stride(0) if mask1 is not None else 0, int(prune_n), int(prune_m),
        err.data_ptr(), err.stride(0), mask_out.data_ptr() + i1 if mask_out is not None else None,
        mask_out.stride(0) if mask_out is not None else 0, _stream()))


def trailing_update(W: torch.Tensor, c0: int, c1: int, err: torch.Tensor, U: torch.Tensor, i1: int, i2: int):
    """`W[:, c0:c1] -= err[:, :i2 - i1] @ U[i1:i2, c0:c1]` in place on fp32 matrix cores (include/vlmc.h: vlmc_sparsegpt_trailing_update):
    the reference's `W[:, i2:] -= Err1.matmul(Hinv[i1:i2, i2:])` (:210) for the columns c0..c1 of it."""
    _need_gpu(W, err, U)
    assert W.dtype == torch.float32 and U.dtype == torch.float32 and err.dtype == torch.float32
    assert W.stride(1) == 1 and U.stride(1) == 1 and err.stride(1) == 1 and 0 < i2 - i1 <= err.shape[1] and err.shape[0] == W.shape[0]
    if c1 <= c0:
        return
    el = W.element_size()
    _lib.check(_lib.load().vlmc_sparsegpt_trailing_update(
        W.data_ptr() + c0 * el, W.shape[0], c1 - c0, W.stride(0), err.data_ptr(), err.stride(0),
        U.data_ptr() + (i1 * U.stride(0) + c0) * el, U.stride(0), i2 - i1, _stream()))


_BLOCK_LOOP = __import__("os").environ.get("VLMC_SGPT_BLOCK_LOOP", "1") != "0"     # 0: the block loop issued from Python, launch by launch (the cross-check)


def prune_blocks(W: torch.Tensor, U: torch.Tensor, blocksize: int, prune_n: int, prune_m: int, rows_per_scope=None, sparsities=None,
                 mask_out: torch.Tensor | None = None):
    """The whole 128-column block loop of `fasterprune` (:167-212) issued by ONE call of the library (include/vlmc.h:
    vlmc_sparsegpt_prune_blocks): sweep + trailing update per block."""
    import ctypes
    _need_gpu(W, U, mask_out)
    assert W.dtype == torch.float32 and U.dtype == torch.float32 and W.stride(1) == 1 and U.stride(1) == 1
    rows, cols = W.shape
    err = torch.empty((rows, min(blocksize, cols)), dtype=torch.float32, device=W.device)
    lib = _lib.load()
    ws, rows_c, sp_c, n = None, None, None, 0
    if prune_n == 0:
        n = len(rows_per_scope)
        assert sum(rows_per_scope) == rows and len(sparsities) == n
        wkey = (W.device.index, torch.cuda.current_stream(W.device).cuda_stream)
        ws = _select_ws.get(wkey)
        if ws is None:
            ws = _select_ws[wkey] = torch.zeros(int(lib.vlmc_sparsegpt_select_workspace_bytes()) // 4, dtype=torch.int32, device=W.device)
        rows_c = (ctypes.c_int64 * n)(*[int(r) for r in rows_per_scope])
        sp_c = (ctypes.c_double * n)(*[float(x) for x in sparsities])
    _lib.check(lib.vlmc_sparsegpt_prune_blocks(
        W.data_ptr(), rows, cols, W.stride(0), U.data_ptr(), U.stride(0), int(blocksize), int(prune_n), int(prune_m), n, rows_c, sp_c,
        err.data_ptr(), err.stride(0), mask_out.data_ptr() if mask_out is not None else None,
        mask_out.stride(0) if mask_out is not None else 0, ws.data_ptr() if ws is not None else None, _stream()))


_SELECT_SWEEP = __import__("os").environ.get("VLMC_SGPT_SELECT_SWEEP", "1") != "0"
_select_ws = {}            # device index -> the zero-filled workspace of vlmc_sparsegpt_select_sweep (returned zero by every call)
_sweep_rows = {}          # device index -> rows the one-launch threshold + sweep takes on that part


def select_sweep_max_rows(device=None):
    """2 workgroups of 32 rows per CU of the device (the C entry point computes the same limit from the CU count and
    returns VLMC_EINVAL above it): 16 384 on an MI355X."""
    idx = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
    if idx not in _sweep_rows:
        _sweep_rows[idx] = 2 * 32 * torch.cuda.get_device_properties(idx).multi_processor_count
    return _sweep_rows[idx]


def select_sweep_usable(rows_per_scope, device=None):
    """The one-launch threshold + sweep takes up to 4 stacked linears and as many rows as 2 x CUs workgroups of 32 hold
    (include/vlmc.h; a scope's last workgroup may be partly empty, hence the margin)."""
    return (_SELECT_SWEEP and 1 <= len(rows_per_scope) <= 4 and all(r > 0 for r in rows_per_scope)
            and sum(rows_per_scope) + 32 * len(rows_per_scope) <= select_sweep_max_rows(device))


def select_sweep_block(W: torch.Tensor, i1: int, i2: int, U: torch.Tensor, rows_per_scope, ranks, err: torch.Tensor,
                       mask_out: torch.Tensor | None = None):
    """Unstructured mode: block threshold (:183-185, per stacked linear) and column sweep (:186-205) of W[:, i1:i2] in ONE
    launch (`vlmc_sparsegpt_select_sweep`); `ranks[s]` = int(rows_s * (i2 - i1) * sparsity_s)."""
    _need_gpu(W, U, err, mask_out)
    assert W.dtype == torch.float32 and U.dtype == torch.float32 and err.dtype == torch.float32
    assert W.stride(1) == 1 and U.stride(1) == 1 and err.stride(1) == 1 and sum(rows_per_scope) == W.shape[0]
    lib = _lib.load()
    wkey = (W.device.index, torch.cuda.current_stream(W.device).cuda_stream)       # sweeps on different streams run side by side
    ws = _select_ws.get(wkey)
    if ws is None:
        ws = _select_ws[wkey] = torch.zeros(int(lib.vlmc_sparsegpt_select_workspace_bytes()) // 4, dtype=torch.int32,
                                            device=W.device)
    import ctypes
    n = len(rows_per_scope)
    rows_c = (ctypes.c_int64 * n)(*[int(r) for r in rows_per_scope])
    ranks_c = (ctypes.c_int64 * n)(*[int(r) for r in ranks])
    count = i2 - i1
    el = W.element_size()
    _lib.check(lib.vlmc_sparsegpt_select_sweep(
        W.data_ptr() + i1 * el, count, W.stride(0), U.data_ptr() + (i1 * U.stride(0) + i1) * el, U.stride(0), n, rows_c, ranks_c,
        err.data_ptr(), err.stride(0), mask_out.data_ptr() + i1 if mask_out is not None else None,
        mask_out.stride(0) if mask_out is not None else 0, ws.data_ptr(), _stream()))


@torch.no_grad()
def fasterprune(layer, H: torch.Tensor, sparsity, prune_n=0, prune_m=0, blocksize=128, percdamp=0.01, return_mask=False,
                factor_cache=None, score_sink=None):
    """`SparseGPT.fasterprune` (:81-215): prunes `layer.weight` in place, sets
    `weight.importance_score`.  H is consumed.  `factor_cache` (a dict owned by the caller, one per distinct
    Hessian) lets linears with the same input reuse the factorization -- bit-identical to recomputing it."""
    W = layer.weight.data.clone().float()
    if factor_cache is not None and "U" in factor_cache:
        U, dead = factor_cache["U"], factor_cache["dead"]
    else:
        U, dead = factorize(H, percdamp, rows_seen=(factor_cache or {}).get("rows_seen"))
        if factor_cache is not None:
            factor_cache["U"], factor_cache["dead"] = U, dead
    W.masked_fill_(dead.unsqueeze(0), 0.0)                                                 # W[:, dead] = 0 (:101)
    diag = torch.diag(U)
    score_mean = (W ** 2 / diag.reshape(1, -1) ** 2).abs().mean()
    rows, cols = W.shape
    pruned = torch.zeros((rows, cols), dtype=torch.bool, device=W.device) if return_mask else None
    fused = prune_n == 0 and _SELECT_THRESHOLD and select_sweep_usable([rows], W.device)
    whole = _BLOCK_LOOP and (prune_n != 0 or fused)
    if whole:                                                                              # :167-212, every block, from one call
        prune_blocks(W, U, blocksize, prune_n, prune_m, [rows], [sparsity], pruned)
    err = None if whole else torch.empty((rows, min(blocksize, cols)), dtype=torch.float32, device=W.device)
    for i1 in (() if whole else range(0, cols, blocksize)):
        i2 = min(i1 + blocksize, cols)
        mask1 = None
        if fused:                                                                          # :183-205 in one launch
            rank = min(int(rows * (i2 - i1) * sparsity), rows * (i2 - i1) - 1)
            select_sweep_block(W, i1, i2, U, [rows], [rank], err, pruned)
            trailing_update(W, i2, cols, err, U, i1, i2)                                   # :210
            continue
        if prune_n == 0:
            tmp = W[:, i1:i2] ** 2 / diag[i1:i2].reshape(1, -1) ** 2                       # :183
            if _SELECT_THRESHOLD:
                # thresh = sort(tmp)[k] is the (k+1)-th smallest score; mask1 = tmp <= thresh = not (tmp > thresh):
                # the multi-tensor radix select (K17) on the one [rows, 128] score block, no sort, no host wait
                keep = ops.score_select(None, "score", scopes=[0], scope_ks=[int(tmp.numel() * sparsity) + 1], scores=[tmp],
                                        apply_weights=False)[0]
                mask1 = torch.logical_not(keep)
            else:
                thresh = torch.sort(tmp.flatten())[0][int(tmp.numel() * sparsity)]         # :184
                mask1 = (tmp <= thresh).contiguous()                                        # :185
        sweep_block(W, i1, i2, U, mask1, prune_n, prune_m, err, pruned)
        trailing_update(W, i2, cols, err, U, i1, i2)                                       # :210
    if score_sink is None:
        setattr(layer.weight, "importance_score", score_mean.item())                       # :165
    else:
        score_sink.append((layer.weight, score_mean))              # read back by the caller, all linears of a block at once
    layer.weight.data = W.reshape(layer.weight.shape).to(layer.weight.data.dtype)          # :215
    return pruned


def stacked_sweeps_enabled():
    """`VLMC_SGPT_STACK=0`: every linear runs its own column sweep (the reference's loop, one linear at a time)."""
    return __import__("os").environ.get("VLMC_SGPT_STACK", "1") != "0"


@torch.no_grad()
def fasterprune_group(layers, sparsities, factor_cache, prune_n=0, prune_m=0, blocksize=128, score_sink=None):
    """`fasterprune` for linears that share ONE factor (q / k / v, wi_0 / wi_1, a cross-attention's k / v: fed the same tensor,
    hence the same Hessian): their weights are stacked along the rows and swept together -- the column sweep, the
    compensation of the columns to the right and the trailing update are row-wise (sparsegpt_pruner.py:186-210), so a
    row gets what its own sweep would give it, while the launches per 128 columns (~11 in unstructured mode) are issued once
    for the group instead of once per linear.  The per-block threshold stays PER LINEAR (`sort(tmp.flatten())[k]` over the
    linear's own [rows, 128] block, :183-185): one multi-tensor radix select with a scope per linear.  `factor_cache` must
    hold the factor (factorize_many)."""
    U, dead = factor_cache["U"], factor_cache["dead"]
    rows = [l.weight.shape[0] for l in layers]
    cols = layers[0].weight.shape[1]
    W = torch.cat([l.weight.data.float() for l in layers], dim=0)
    W.masked_fill_(dead.unsqueeze(0), 0.0)                                                 # W[:, dead] = 0 (:101)
    diag = torch.diag(U)
    dsq = diag.reshape(1, -1) ** 2
    bounds = [0]
    for r in rows:
        bounds.append(bounds[-1] + r)
    means = [(W[bounds[i]:bounds[i + 1]] ** 2 / dsq).abs().mean() for i in range(len(layers))]
    fused = prune_n == 0 and _SELECT_THRESHOLD and select_sweep_usable(rows, W.device)
    whole = _BLOCK_LOOP and (prune_n != 0 or fused)
    if whole:                                                                              # :167-212, every block, from one call
        prune_blocks(W, U, blocksize, prune_n, prune_m, rows, sparsities, None)
    keep = torch.empty((W.shape[0], min(blocksize, cols)), dtype=torch.bool, device=W.device) if (prune_n == 0 and not whole) else None
    err = None if whole else torch.empty((W.shape[0], min(blocksize, cols)), dtype=torch.float32, device=W.device)
    for i1 in (() if whole else range(0, cols, blocksize)):
        i2 = min(i1 + blocksize, cols)
        mask1 = None
        if fused:                                                                          # :183-205 in one launch, a scope per linear
            ranks = [min(int(r * (i2 - i1) * sp), r * (i2 - i1) - 1) for r, sp in zip(rows, sparsities)]
            select_sweep_block(W, i1, i2, U, rows, ranks, err, None)
            trailing_update(W, i2, cols, err, U, i1, i2)                                   # :210
            continue
        if prune_n == 0:
            tmp = W[:, i1:i2] ** 2 / dsq[:, i1:i2]                                          # :183
            kb = keep if i2 - i1 == keep.shape[1] else torch.empty((W.shape[0], i2 - i1), dtype=torch.bool, device=W.device)
            parts = [tmp[bounds[i]:bounds[i + 1]] for i in range(len(layers))]
            ops.score_select(None, "score", scopes=list(range(len(layers))),
                             scope_ks=[int(p.numel() * sp) + 1 for p, sp in zip(parts, sparsities)], scores=parts,
                             apply_weights=False, keeps=[kb[bounds[i]:bounds[i + 1]] for i in range(len(layers))])
            mask1 = torch.logical_not(kb)
        sweep_block(W, i1, i2, U, mask1, prune_n, prune_m, err, None)
        trailing_update(W, i2, cols, err, U, i1, i2)                                       # :210
    for i, layer in enumerate(layers):
        if score_sink is None:
            setattr(layer.weight, "importance_score", means[i].item())                     # :165
        else:
            score_sink.append((layer.weight, means[i]))
        layer.weight.data = W[bounds[i]:bounds[i + 1]].reshape(layer.weight.shape).to(layer.weight.data.dtype)   # :215


def flush_scores(score_sink):
    """`weight.importance_score` (a Python float, :165) for everything `fasterprune(..., score_sink=...)` has queued: one
    host copy."""
    if score_sink:
        vals = torch.stack([s for _, s in score_sink]).cpu().tolist()
        for (w, _), v in zip(score_sink, vals):
            setattr(w, "importance_score", v)
        score_sink.clear()
