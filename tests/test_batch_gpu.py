"""GPU: the batched C-ABI entry points (all hook inputs / statistics / linears of one transformer
block per launch) against the single-job entry points and the CPU oracle, bit for bit; plus the
SEL_MATRIX selection's fast path (sample bracket + two counting passes) and its exact fallback."""
import numpy as np
import pytest
import torch

from oracle import wanda as OW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from vlmc import ops
    return ops


def _w(out_f, in_f, dtype, seed, zero_frac=0.0):
    g = torch.Generator().manual_seed(seed)
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(dtype)
    if zero_frac:
        W[torch.rand(out_f, in_f, generator=g) < zero_frac] = 0
    s = (torch.rand(in_f, generator=g) * 4 + 0.01).numpy().astype(np.float32)
    return W, s


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_sqnorm_batch_equals_single_calls_and_oracle(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    shapes = [(5, 37, 1408), (5, 37, 6144), (5, 9, 2048), (5, 37, 100), (5, 1, 8)]
    xs = [((torch.randn(s, generator=g) * 2) + 0.1).to(dtype).to(DEV) for s in shapes]
    flat = torch.full((5, sum(s[2] for s in shapes) + 3), -1.0, dtype=torch.float32, device=DEV)
    outs, off = [], 0
    for s in shapes:
        outs.append(flat[:, off:off + s[2]])
        off += s[2]
    ops.act_sqnorm_batch(xs, outs)
    for x, o in zip(xs, outs):
        single = ops.act_sqnorm(x)
        assert torch.equal(o, single)
        want = np.stack([OW.act_sqnorm(x[c].cpu()) for c in range(x.shape[0])])
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert bool((flat[:, off:] == -1.0).all())                   # nothing written past the jobs' columns
    auto = ops.act_sqnorm_batch(xs)                              # library-allocated outputs
    assert all(torch.equal(a, o) for a, o in zip(auto, outs))


def test_sqnorm_batch_more_jobs_than_one_launch_holds():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    xs = [torch.randn((3, 5, 64 + 8 * j), generator=g).to(torch.bfloat16).to(DEV) for j in range(15)]
    outs = ops.act_sqnorm_batch(xs)
    for x, o in zip(xs, outs):
        assert torch.equal(o, ops.act_sqnorm(x))


def test_scaler_update_batch_equals_single_calls():
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    widths = [1408, 6144, 2048, 100, 8]
    calls = 130                                                  # > one LDS panel of 128 calls
    flat = (torch.rand((calls, sum(widths)), generator=g) * 50).to(DEV)
    nsqs, off = [], 0
    for w in widths:
        nsqs.append(flat[:, off:off + w])
        off += w
    scal = [torch.zeros(w, dtype=torch.float32, device=DEV) for w in widths]
    sq = [torch.empty(w, dtype=torch.float32, device=DEV) for w in widths]
    n = ops.wanda_scaler_update_batch(scal, 0, nsqs, 1, sq)
    assert n == calls
    for w, nsq, s_b, q_b in zip(widths, nsqs, scal, sq):
        s1 = torch.zeros(w, dtype=torch.float32, device=DEV)
        q1 = torch.empty_like(s1)
        ops.wanda_scaler_update(s1, 0, nsq.contiguous(), 1, sqrt_out=q1)
        assert torch.equal(s1, s_b) and torch.equal(q1, q_b)
        ref, nn = np.zeros(w, np.float32), 0
        for c in range(calls):
            ref, nn = OW.scaler_update(ref, nn, nsq[c].cpu().numpy(), 1)
        assert np.array_equal(ref.view(np.uint32), s_b.cpu().numpy().view(np.uint32))
    # continuing from a non-zero sample count with batch 2
    n2 = ops.wanda_scaler_update_batch(scal[:2], n, [nsqs[0][:3], nsqs[1][:3]], 2, None)
    assert n2 == calls + 6


def _single(W, s, mode, **kw):
    ops = _ops()
    Wd = W.clone().to(DEV)
    mask, parts = ops.wanda_select(Wd, ops.sqrt_scaler(torch.from_numpy(s).to(DEV)), mode, **kw)
    return mask, Wd, parts


@pytest.mark.parametrize("mode", ["row", "matrix", "nm"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_select_batch_equals_single_calls(mode, dtype):
    ops = _ops()
    shapes = [(64, 256), (64, 256), (96, 512), (40, 256), (33, 104), (96, 512), (8, 256)]
    cases = [_w(o, i, dtype, 100 + j, zero_frac=0.1 if j == 3 else 0.0) for j, (o, i) in enumerate(shapes)]
    Ws = [c[0].clone().to(DEV) for c in cases]
    sqs = [ops.sqrt_scaler(torch.from_numpy(c[1]).to(DEV)) for c in cases]
    if mode == "row":
        kw, ks = {}, [int(i * 0.5) for _, i in shapes]
    elif mode == "matrix":
        kw, ks = {}, [int(o * i * 0.45) for o, i in shapes]
    else:
        kw, ks = dict(n=2, m=4), None
    masks, parts = ops.wanda_select_batch(Ws, sqs, mode, ks=ks, apply_zero=True, **kw)
    for j, (W, s) in enumerate(cases):
        skw = dict(kw) if mode == "nm" else dict(k=ks[j])
        m1, W1, p1 = _single(W, s, mode, **skw)
        assert torch.equal(masks[j], m1), (mode, j)
        assert torch.equal(Ws[j], W1), (mode, j)
        assert float(parts[j].sum()) == pytest.approx(float(p1.sum()), rel=1e-12)
        want = OW.prune_linear(W, s, mode, ratio=0.5 if mode == "row" else 0.45, n=2 if mode == "nm" else 0,
                               m=4 if mode == "nm" else 0)
        assert np.array_equal(masks[j].cpu().numpy(), want["mask"]), (mode, j)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shapes", [
    [(64, 2048), (40, 5120), (33, 2048), (6, 2048)],                       # T5-XL widths: 1 wave / 4 waves x 3 chunks
    [(16, 4096), (7, 8192), (5, 1024), (9, 2048), (3, 6144), (21, 520)],    # 2 / 4 / 3 chunks per lane, short narrow rows
])
def test_select_batch_mixed_row_widths(dtype, shapes):
    """Narrow (<= 2048 columns) and wide rows of one call share ONE launch (select_rows_mixed_kernel):
    results must equal the single calls (the tuned per-width kernels) and the oracle, ties and k = 0 / in included."""
    ops = _ops()
    cases = [_w(o, i, dtype, 300 + j, zero_frac=0.6 if j == 1 else (0.2 if j == 2 else 0.0)) for j, (o, i) in enumerate(shapes)]
    Ws = [c[0].clone().to(DEV) for c in cases]
    sqs = [ops.sqrt_scaler(torch.from_numpy(c[1]).to(DEV)) for c in cases]
    for ratios in ([0.5] * len(shapes), [0.3, 0.7, 0.0, 1.0, 0.5, 0.9][:len(shapes)]):
        for W, c in zip(Ws, cases):
            W.copy_(c[0])
        ks = [int(i * r) for (_, i), r in zip(shapes, ratios)]
        masks, parts = ops.wanda_select_batch(Ws, sqs, "row", ks=ks, apply_zero=True)
        for j, (W, s) in enumerate(cases):
            m1, W1, p1 = _single(W, s, "row", k=ks[j])
            assert torch.equal(masks[j], m1), j
            assert torch.equal(Ws[j], W1), j
            # fp32 lane sums grouped by another number of waves per row: the importance score's 1e-5 contract
            assert float(parts[j].sum()) == pytest.approx(float(p1.sum()), rel=1e-6)
            assert int((~masks[j]).sum()) == ks[j] * shapes[j][0]
            score = OW.wanda_score(W, s)
            assert np.array_equal(~masks[j].cpu().numpy(), OW.select_rows(score, ks[j])), j


def _check_matrix(W, s, k):
    mask, Wd, parts = _single(W, s, "matrix", k=k)
    score = OW.wanda_score(W, s)
    pruned = OW.select_matrix(score, k)
    assert np.array_equal(mask.cpu().numpy(), ~pruned)
    want_w = W.clone()
    want_w[torch.from_numpy(pruned)] = 0
    assert torch.equal(Wd.cpu(), want_w)
    return int(pruned.sum())


@pytest.mark.parametrize("shape", [(4224, 1408), (1408, 6144), (1000, 1001)])
@pytest.mark.parametrize("ratio", [0.5, 0.02, 0.97])
def test_matrix_select_model_shapes_vs_oracle(shape, ratio):
    W, s = _w(shape[0], shape[1], torch.float16, 11)
    n = _check_matrix(W, s, int(W.numel() * ratio))
    assert abs(n - int(W.numel() * ratio)) <= 64                 # strict '<' may drop a few tied elements only


@pytest.mark.parametrize("when", ["1", "2"])
def test_matrix_select_fallback_path_is_exact(when, monkeypatch):
    """1: the job fails before anything is written.  2: it fails at barrier B, AFTER the chunks that the merged histogram
    decides have been written (mask bytes and zeroed weights): the exact select of the job's last workgroup, reading a W
    that is already partly zeroed, must still write the same mask and weights."""
    monkeypatch.setenv("VLMC_MATRIX_FORCE_SLOW", when)
    W, s = _w(512, 1408, torch.float16, 12)
    _check_matrix(W, s, int(W.numel() * 0.5))
    _check_matrix(W, s, 0)
    _check_matrix(W, s, W.numel() - 1)
    for seed, shape, ratio in ((41, (4224, 1408), 0.5), (42, (1408, 6144), 0.3), (43, (1000, 1001), 0.7)):
        W, s = _w(shape[0], shape[1], torch.float16, seed)
        _check_matrix(W, s, int(W.numel() * ratio))
    W, s = _w(600, 1408, torch.bfloat16, 44, zero_frac=0.5)          # re-pruning half-zero weights
    _check_matrix(W, s, int(W.numel() * 0.6))


@pytest.mark.parametrize("zero_frac,ratio", [(0.5, 0.5), (0.5, 0.25), (0.5, 0.75), (0.9, 0.95)])
def test_matrix_select_already_sparse_weights(zero_frac, ratio):
    """Half the scores are exactly 0: the sample bracket spans from key 0 to a normal float (wider than
    the two counting passes resolve) or sits inside the tie run -- both must still be exact."""
    W, s = _w(600, 1408, torch.bfloat16, 13, zero_frac=zero_frac)
    _check_matrix(W, s, int(W.numel() * ratio))


def test_matrix_select_extreme_ranks_and_nan():
    W, s = _w(300, 704, torch.float16, 14)
    for k in (0, 1, W.numel() // 2, W.numel() - 2, W.numel() - 1):
        _check_matrix(W, s, k)
    s2 = s.copy()
    s2[5] = np.nan                                               # NaN scores sort last; a NaN threshold prunes nothing
    mask, _, _ = _single(W, s2, "matrix", k=W.numel() - 1)
    assert bool(mask.all())
    _check_matrix(W, s2, W.numel() // 3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_matrix_select_fused_streams_what_the_registers_do_not_hold(monkeypatch, dtype):
    """With a grid of 8 workgroups a lane's register chunks cover only part of a 1408 x 1408 matrix: the rest is
    streamed in the count pass and re-read in the candidate and apply passes -- same exact result."""
    monkeypatch.setenv("VLMC_MATRIX_FUSED_WGS", "8")
    for seed, shape in ((21, (1408, 1408)), (22, (700, 1001))):          # aligned and unaligned rows
        W, s = _w(shape[0], shape[1], dtype, seed)
        for ratio in (0.5, 0.1):
            _check_matrix(W, s, int(W.numel() * ratio))


def test_matrix_select_fused_equals_four_launch_form(monkeypatch):
    """VLMC_MATRIX_FUSED=0 (sample / count / apply / resolve launches) is the cross-check of the fused kernel; rows wider
    than the fused kernel's LDS table (in > 8192) always take it.  Both leave the workspace zero-filled for the other."""
    ops = _ops()
    shapes = [(1408, 1408), (64, 8200), (300, 6144)]
    cases = [_w(o, i, torch.float16, 500 + j) for j, (o, i) in enumerate(shapes)]
    sqs = [ops.sqrt_scaler(torch.from_numpy(c[1]).to(DEV)) for c in cases]
    ks = [int(o * i * 0.4) for o, i in shapes]
    got = {}
    for fused in ("1", "0", "1"):
        monkeypatch.setenv("VLMC_MATRIX_FUSED", fused)
        Ws = [c[0].clone().to(DEV) for c in cases]
        masks, parts = ops.wanda_select_batch(Ws, sqs, "matrix", ks=ks, apply_zero=True)
        cur = ([m.clone() for m in masks], [w.clone() for w in Ws])
        if got:
            assert all(torch.equal(a, b) for a, b in zip(got["m"], cur[0]))
            assert all(torch.equal(a, b) for a, b in zip(got["w"], cur[1]))
        got = {"m": cur[0], "w": cur[1]}
    for (W, s), mk, k in zip(cases, got["m"], ks):
        assert np.array_equal(mk.cpu().numpy(), ~OW.select_matrix(OW.wanda_score(W, s), k))


@pytest.mark.parametrize("zero_frac,ratio", [(0.5, 0.25), (0.5, 0.4999), (0.97, 0.5)])
def test_matrix_select_tie_floods_at_model_size(zero_frac, ratio):
    """The threshold falls inside a run of millions of equal scores (weights pruned before, dead channels): the fused
    kernel refines the crowded bin from its registers down to one key value (no candidates, no streaming fallback)."""
    W, s = _w(6144, 1408, torch.float16, 31, zero_frac=zero_frac)
    _check_matrix(W, s, int(W.numel() * ratio))


def test_matrix_select_two_streams_compete_for_the_cus():
    """The fused matrix-wide kernel wants one workgroup per CU resident at the same time.  Two streams launching
    it concurrently can each get only part of the chip: the bounded barrier wait must then run out and hand the
    job to the exact streaming fallback (slow, never a hang, never a wrong mask)."""
    ops = _ops()
    shapes = [(1408, 1408), (1408, 6144), (4224, 1408)]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    runs = []
    for si, st in enumerate(streams):
        cases = [_w(o, i, torch.float16, 400 + 10 * si + j) for j, (o, i) in enumerate(shapes)]
        with torch.cuda.stream(st):
            Ws = [c[0].clone().to(DEV) for c in cases]
            sqs = [ops.sqrt_scaler(torch.from_numpy(c[1]).to(DEV)) for c in cases]
            masks = [torch.empty(w.shape, dtype=torch.bool, device=DEV) for w in Ws]
            parts = [torch.empty(ops.select_partials("matrix", *w.shape), dtype=torch.float64, device=DEV) for w in Ws]
            ks = [int(w.numel() * 0.5) for w in Ws]
            plan = ops.plan_select_batch(Ws, sqs, "matrix", ks=ks, apply_zero=True, masks=masks, partials=parts)
        runs.append((cases, Ws, masks, ks, plan))
    torch.cuda.synchronize()
    for rep in range(3):                                        # (re-pruning pruned weights: heavy ties as well)
        for (cases, Ws, masks, ks, plan), st in zip(runs, streams):
            with torch.cuda.stream(st):
                if rep < 2:
                    for W, c in zip(Ws, cases):
                        W.copy_(c[0], non_blocking=True)
                plan()
        torch.cuda.synchronize()
        if rep < 2:
            for cases, Ws, masks, ks, plan in runs:
                for (W, s), Wd, mk, k in zip(cases, Ws, masks, ks):
                    pruned = OW.select_matrix(OW.wanda_score(W, s), k)
                    assert np.array_equal(mk.cpu().numpy(), ~pruned)
                    want = W.clone()
                    want[torch.from_numpy(pruned)] = 0
                    assert torch.equal(Wd.cpu(), want)


def test_select_batch_rejects_shared_workspace_and_bad_args():
    from vlmc import _lib
    ops = _ops()
    W, s = _w(16, 64, torch.float16, 15)
    Wd = W.to(DEV)
    sq = ops.sqrt_scaler(torch.from_numpy(s).to(DEV))
    mk = torch.empty((16, 64), dtype=torch.bool, device=DEV)
    ws = torch.empty(1 << 18, dtype=torch.uint8, device=DEV)
    jobs = (_lib.SelectJob * 2)()
    for j in range(2):
        jobs[j] = _lib.SelectJob(Wd.data_ptr(), 16, 64, 64, sq.data_ptr(), 10, mk.data_ptr(), None, ws.data_ptr(), ws.numel())
    lib = _lib.load()
    assert lib.vlmc_wanda_select_batch(jobs, 2, _lib.F16, _lib.SEL_MATRIX, 0, 0, 1, None) == _lib.VLMC_EINVAL
    assert b"share a workspace" in lib.vlmc_last_error()
    assert lib.vlmc_wanda_select_batch(jobs, 0, _lib.F16, _lib.SEL_ROW, 0, 0, 1, None) == _lib.VLMC_EINVAL
