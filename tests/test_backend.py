"""Feature back-end.  CPU part: the oracle restatement (oracle/orc_backend.c, parity unpinned
against the reference -- see that file) against independent numpy formulations of the same definitions.  GPU part: the
HIP kernels against the oracle, bit for bit (both follow the reference's operation order)."""
import numpy as np
import pytest

from oracle.binding import oracle_matrix_multiply, oracle_normalize, oracle_regression


def seg(n, dim, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    return (r.standard_normal((n, dim)) * 3 + r.standard_normal(dim) * 5).astype(np.float32)


# ------------------------------------------------------------------ oracle vs the definitions
@pytest.mark.parametrize("variance", [False, True])
def test_oracle_whole_segment_normalisation(variance):
    x = seg(500, 16, 1)
    got = oracle_normalize(x, variance=variance)
    mean = (x.astype(np.float64).sum(0) / len(x)).astype(np.float32)
    want = x - mean
    if variance:
        s = x.astype(np.float64)
        sd = np.sqrt(((s * s).sum(0) - s.sum(0) ** 2 / len(x)) / len(x)).astype(np.float32)
        want = want / sd
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
    assert np.allclose(got.mean(0), 0, atol=1e-4)
    if variance:
        assert np.allclose(got.std(0), 1, atol=1e-3)
    const = np.full((7, 3), 2.5, np.float32)          # zero variance -> standard deviation forced to 1
    assert np.array_equal(oracle_normalize(const, variance=True), np.zeros((7, 3), np.float32))


def test_oracle_sliding_window_normalisation():
    """length 21, right 10: frame u is normalised with the frames [u - 10, u + 10] that exist; the last `right` frames keep
    the statistics of the last full step (Normalization::update does not update them while flushing)"""
    L, R, n = 21, 10, 60
    x = seg(n, 4, 2)
    got = oracle_normalize(x, length=L, right=R)
    for u in range(n):
        hi = u + R if u + R <= n - 1 else n - 1
        lo = max(0, hi - L + 1)
        want = x[u] - x[lo:hi + 1].astype(np.float64).mean(0).astype(np.float32)
        assert np.allclose(got[u], want, rtol=1e-5, atol=1e-5), u
    short = seg(6, 4, 3)                               # fewer frames than `right`: everything leaves at flush time
    assert np.allclose(oracle_normalize(short, length=L, right=R), short - short.mean(0), atol=1e-5)


def test_oracle_regression_is_the_least_squares_slope():
    n, right = 40, 2
    x = seg(n, 5, 4)
    d1 = oracle_regression(x, 1, right)
    d2 = oracle_regression(x, 2, right)
    tt = np.arange(-right, right + 1, dtype=np.float64)
    for t in range(n):
        idx = np.clip(np.arange(t - right, t + right + 1), 0, n - 1)     # copy margin
        w = x[idx].astype(np.float64)
        slope = (tt[:, None] * w).sum(0) / (tt ** 2).sum()
        assert np.allclose(d1[t], slope, rtol=1e-5, atol=1e-5)
        quad = np.polyfit(tt, w, 2)[0] * 2                                # second derivative of the LS parabola
        assert np.allclose(d2[t], quad, rtol=1e-4, atol=1e-4)
    line = (np.arange(n, dtype=np.float32)[:, None] * np.float32(0.5)) + np.float32(3)
    assert np.allclose(oracle_regression(line, 1, 2)[2:-2], 0.5, atol=1e-6)
    assert np.allclose(oracle_regression(line, 2, 2)[2:-2], 0.0, atol=1e-5)


def test_oracle_matrix_multiply():
    r = np.random.Generator(np.random.PCG64(5))
    M = r.standard_normal((9, 33)).astype(np.float32)
    x = r.standard_normal((20, 33)).astype(np.float32)
    assert np.allclose(oracle_matrix_multiply(M, x), x.astype(np.float64) @ M.T.astype(np.float64), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ HIP kernels vs the oracle
def _plan(ctx, lens):
    """an MFCC plan whose segments have exactly `lens` frames (frame i covers samples [160 i, 160 i + 400))"""
    import rasr_amd
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=8)
    samples = [400 + 160 * (n - 1) for n in lens]
    off = np.concatenate([[0], np.cumsum(samples)])
    plan = fe.plan(off)
    assert list(np.diff(plan.frame_offsets)) == list(lens)
    return plan


@pytest.mark.gpu
def test_backend_kernels_match_the_oracle(ctx):
    import torch
    lens = [1, 2, 5, 9, 64, 333, 7]
    plan = _plan(ctx, lens)
    F, dim, ld = sum(lens), 13, 20
    x = np.zeros((F, ld), np.float32)
    x[:, 3:3 + dim] = seg(F, dim, 11)
    xd = torch.from_numpy(x).cuda()
    ctx.use_torch_stream()
    off = np.concatenate([[0], np.cumsum(lens)])
    view = lambda a: [a[off[i]:off[i + 1], 3:3 + dim] for i in range(len(lens))]
    for kw in (dict(), dict(variance=True), dict(length=21, right=10), dict(variance=True, length=5, right=0), dict(length=9, right=8)):
        out = torch.full((F, 16), 7.0, dtype=torch.float32, device="cuda")
        ctx.normalize(plan, xd[:, 3:], ld, dim, out[:, 1:], 16, **kw)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.all(got[:, 0] == 7.0) and np.all(got[:, 1 + dim:] == 7.0)          # only the addressed columns are written
        for i, s in enumerate(view(x)):
            want = oracle_normalize(s, **kw)
            assert np.array_equal(got[off[i]:off[i + 1], 1:1 + dim].view(np.uint32), want.view(np.uint32)), (kw, i)
    for order in (1, 2):
        for right in (1, 2, 4):
            out = torch.zeros((F, dim), dtype=torch.float32, device="cuda")
            ctx.regression(plan, xd[:, 3:], ld, dim, out, dim, order=order, right=right)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            for i, s in enumerate(view(x)):
                want = oracle_regression(s, order, right)
                assert np.array_equal(got[off[i]:off[i + 1]].view(np.uint32), want.view(np.uint32)), (order, right, i)
    M = np.random.Generator(np.random.PCG64(12)).standard_normal((45, dim)).astype(np.float32)
    Md = torch.from_numpy(M).cuda()
    out = torch.zeros((F, 48), dtype=torch.float32, device="cuda")
    ctx.matrix_multiply(Md, 45, dim, xd[:, 3:], ld, F, out, 48)
    torch.cuda.synchronize()
    want = oracle_matrix_multiply(M, x[:, 3:3 + dim])
    assert np.array_equal(out.cpu().numpy()[:, :45].view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_backend_chain_cmvn_derivatives_lda(ctx):
    """processing.standard_system.flow in one pass: MFCC-16 -> segment CMVN -> [c, delta, deltadelta(c0)] -> 3-frame window
    -> LDA, all device resident; the result equals the oracle's chain on every segment"""
    import torch

    import rasr_amd
    from oracle import OracleMfcc
    from tests import synth
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=16)
    lens = [16000, 5000, 23000]
    pcm = np.concatenate([synth.waveform(n, seed=30 + i) for i, n in enumerate(lens)])
    plan = fe.plan(np.concatenate([[0], np.cumsum(lens)]))
    F = plan.total_frames
    ctx.use_torch_stream()
    ceps = torch.empty((F, 16), dtype=torch.float32, device="cuda")
    fe.run_plan(plan, torch.from_numpy(pcm).cuda(), ceps)
    feat = torch.zeros((F, 33), dtype=torch.float32, device="cuda")       # [16 normalised | 16 delta | 1 deltadelta of c0]
    ctx.normalize(plan, ceps, 16, 16, feat, 33)
    ctx.regression(plan, feat, 33, 16, feat[:, 16:], 33, order=1, right=2)
    ctx.regression(plan, feat, 33, 1, feat[:, 32:], 33, order=2, right=2)
    win = torch.empty((F, 99), dtype=torch.float32, device="cuda")
    ctx.context_window(plan, feat, 33, 1, 1, win, 99)
    M = np.random.Generator(np.random.PCG64(40)).standard_normal((24, 99)).astype(np.float32)
    out = torch.empty((F, 24), dtype=torch.float32, device="cuda")
    ctx.matrix_multiply(torch.from_numpy(M).cuda(), 24, 99, win, 99, F, out, 24)
    torch.cuda.synchronize()
    got, c = out.cpu().numpy(), ceps.cpu().numpy()
    fo = plan.frame_offsets
    for i in range(len(lens)):
        s = c[fo[i]:fo[i + 1]]
        nrm = oracle_normalize(s)
        f = np.concatenate([nrm, oracle_regression(nrm, 1, 2), oracle_regression(nrm[:, :1], 2, 2)], axis=1)
        n = len(f)
        w = np.concatenate([f[np.clip(np.arange(n) + k, 0, n - 1)] for k in (-1, 0, 1)], axis=1)
        assert np.array_equal(got[fo[i]:fo[i + 1]].view(np.uint32), oracle_matrix_multiply(M, w).view(np.uint32)), i


@pytest.mark.gpu
def test_overlapping_views_are_rejected(ctx):
    """sliding-window normalisation, regression and the matrix product read neighbouring rows while others are written: views
    that share memory are refused (disjoint column ranges of one wide matrix are not); whole-segment normalisation in place on
    the identical view stays allowed and equals the out-of-place result"""
    import torch

    import rasr_amd
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=12)
    plan = fe.plan(np.array([0, 16000, 40000], np.int64))
    T = plan.total_frames
    wide = torch.randn((T, 40), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    a, b = wide[:, 0:12], wide[:, 12:24]
    ctx.normalize(plan, a, 40, 12, b, 40, variance=False, length=20, right=5)      # disjoint columns of one matrix: fine
    ctx.regression(plan, a, 40, 12, b, 40, order=1, right=2)
    shifted = wide[:, 6:18]
    for call in (lambda: ctx.normalize(plan, a, 40, 12, shifted, 40, length=20, right=5),
                 lambda: ctx.normalize(plan, a, 40, 12, a, 40, length=20, right=5),
                 lambda: ctx.regression(plan, a, 40, 12, a, 40, order=1, right=2),
                 lambda: ctx.regression(plan, a, 40, 12, shifted, 40, order=2, right=2),
                 lambda: ctx.matrix_multiply(wide[:12, :12].contiguous(), 12, 12, a, 40, T, shifted, 40)):
        with pytest.raises(rasr_amd.AmxError, match="overlap"):
            call()
    ref = torch.empty((T, 12), dtype=torch.float32, device="cuda")
    src = a.contiguous()
    ctx.normalize(plan, src, 12, 12, ref, 12, variance=True)
    ctx.normalize(plan, src, 12, 12, src, 12, variance=True)                        # whole segment, in place
    torch.cuda.synchronize()
    assert torch.equal(src, ref)


# ------------------------------------------------------------------ the node's other types: divide-by-mean, level, mean-and-variance-1D
def test_oracle_other_normalisation_types_against_definitions():
    from oracle.binding import oracle_normalize_ex
    x = np.abs(seg(300, 6, 21)) + 1
    got = oracle_normalize_ex(x, 2)                                     # divide-by-mean, whole segment
    assert np.allclose(got, x / (x.astype(np.float64).sum(0) / len(x)).astype(np.float32), rtol=1e-6)
    got = oracle_normalize_ex(x, 3, level=2)                            # level: component 2 minus its maximum, rest untouched
    want = x.copy()
    want[:, 2] -= x[:, 2].max()
    assert np.array_equal(got, want) and got[:, 2].max() == 0
    got = oracle_normalize_ex(x, 4)                                     # one mean / deviation over everything
    flat = x.astype(np.float64)
    assert np.allclose(got, (x - np.float32(flat.mean())) / np.float32(flat.std()), rtol=1e-5, atol=1e-6)
    L, R, n = 11, 4, 40                                                 # sliding level: max over the frames [u + R - L + 1, u + R]
    got = oracle_normalize_ex(x[:n], 3, level=0, length=L, right=R)
    for u in range(n):
        hi = min(u + R, n - 1)
        lo = max(0, hi - L + 1)
        assert got[u, 0] == np.float32(x[u, 0] - x[lo:hi + 1, 0].max()), u
    const = np.full((5, 3), 2.0, np.float32)
    assert np.array_equal(oracle_normalize_ex(const, 4), np.zeros((5, 3), np.float32))      # zero deviation -> 1


@pytest.mark.gpu
def test_other_normalisation_types_match_the_oracle(ctx):
    import torch

    from oracle.binding import oracle_normalize_ex
    lens = [1, 2, 5, 9, 64, 333, 7]
    plan = _plan(ctx, lens)
    F, dim, ld = sum(lens), 13, 20
    x = np.zeros((F, ld), np.float32)
    x[:, 3:3 + dim] = np.abs(seg(F, dim, 31)) + 0.5
    xd = torch.from_numpy(x).cuda()
    ctx.use_torch_stream()
    off = np.concatenate([[0], np.cumsum(lens)])
    for type_, level in ((2, 0), (3, 0), (3, 12), (4, 0)):
        for length, right in ((0, 0), (21, 10), (5, 0), (9, 8)):
            out = torch.full((F, 16), 7.0, dtype=torch.float32, device="cuda")
            ctx.normalize_ex(plan, xd[:, 3:], ld, dim, out[:, 1:], 16, type_, level=level, length=length, right=right)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            assert np.all(got[:, 0] == 7.0) and np.all(got[:, 1 + dim:] == 7.0)
            for i in range(len(lens)):
                s = x[off[i]:off[i + 1], 3:3 + dim]
                want = oracle_normalize_ex(s, type_, level=level, length=length, right=right)
                assert np.array_equal(got[off[i]:off[i + 1], 1:1 + dim].view(np.uint32), want.view(np.uint32)), (type_, level, length, right, i)
    import rasr_amd
    out = torch.zeros((F, dim), dtype=torch.float32, device="cuda")
    with pytest.raises(rasr_amd.AmxError, match="level index"):
        ctx.normalize_ex(plan, xd[:, 3:], ld, dim, out, dim, 3, level=dim)
    with pytest.raises(rasr_amd.AmxError, match="unknown type"):
        ctx.normalize_ex(plan, xd[:, 3:], ld, dim, out, dim, 9)


# ------------------------------------------------------------------ signal-vector-f32-*-normalization (per-vector)
VNORM = ["amplitude-spectrum-energy", "energy", "maximum", "mean-energy", "mean", "variance"]


def test_oracle_vector_normalisations_against_definitions():
    """oracle/orc_backend.c: orc_vector_normalize against the node names' definitions: unit (Parseval / plain / mean) energy, unit
    maximum, zero mean, zero mean and unit deviation"""
    from oracle.binding import oracle_vector_normalize
    x = np.abs(seg(40, 257, 31)) + 0.1
    x64 = x.astype(np.float64)
    y = oracle_vector_normalize(x, "energy").astype(np.float64)
    assert np.allclose((y * y).sum(1), 1, atol=1e-5)
    y = oracle_vector_normalize(x, "mean-energy").astype(np.float64)
    assert np.allclose((y * y).mean(1), 1, atol=1e-5)
    y = oracle_vector_normalize(x, "amplitude-spectrum-energy").astype(np.float64)
    e = (y[:, 0] ** 2 + y[:, -1] ** 2 + 2 * (y[:, 1:-1] ** 2).sum(1)) / (2 * 256)
    assert np.allclose(e, 1, atol=1e-5)
    assert np.allclose(oracle_vector_normalize(x, "maximum").max(1), 1, atol=1e-6)
    assert np.allclose(oracle_vector_normalize(x, "mean"), x64 - x64.mean(1, keepdims=True), atol=1e-5)
    y = oracle_vector_normalize(x, "variance").astype(np.float64)
    assert np.allclose(y.mean(1), 0, atol=1e-5) and np.allclose(y.std(1), 1, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", VNORM)
def test_vector_normalisations_match_the_oracle(ctx, kind):
    """the kernel follows the same operation order (f32 products added to a double in index order, f32 scaling): bit-identical, with
    row strides, in place on the identical view, a zero vector (division by zero like the reference: inf / NaN) and dim 2"""
    import torch

    import rasr_amd
    from oracle.binding import oracle_vector_normalize
    ctx.use_torch_stream()
    for n, dim in ((1, 2), (300, 40), (77, 257), (1000, 13)):
        x = seg(n, dim, 40 + dim)
        if n > 5:
            x[3] = 0.0
        want = oracle_vector_normalize(x, kind)
        wide = torch.zeros((n, dim + 5), dtype=torch.float32, device="cuda")
        wide[:, 2:2 + dim] = torch.from_numpy(x).cuda()
        out = torch.full((n, dim + 3), 9.0, dtype=torch.float32, device="cuda")
        ctx.vector_normalize(kind, wide[:, 2:], dim + 5, n, dim, out[:, 1:], dim + 3)
        torch.cuda.synchronize()
        got = out[:, 1:1 + dim].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kind, n, dim)
        assert (out[:, 0] == 9.0).all() and (out[:, dim + 1:] == 9.0).all()
        ctx.vector_normalize(kind, wide[:, 2:], dim + 5, n, dim, wide[:, 2:], dim + 5)        # in place
        torch.cuda.synchronize()
        assert np.array_equal(wide[:, 2:2 + dim].cpu().numpy().view(np.uint32), want.view(np.uint32))
    with pytest.raises(rasr_amd.AmxError, match="overlap"):
        ctx.vector_normalize(kind, wide[:, 2:], dim + 5, n, dim, wide[:, 3:], dim + 5)


# ------------------------------------------------------------------ generic-vector-f32-<function> (Flow/SimpleFunction.hh)
VFUNC_EXACT = {"power": 0.33, "sqrt": 0.0, "addition": -3.25, "multiplication": 500.0, "quantize": 1.0, "abs": 0.0, "minimum": 0.5, "maximum": 0.5}
VFUNC_ULPS = {"log": 0.0, "log-plus": 1.5, "ln": 0.0, "exp": 0.0, "cos": 0.0}


def test_oracle_vector_functions_against_definitions():
    from oracle.binding import oracle_vector_function
    import rasr_amd
    K = rasr_amd.Context.VECTOR_FUNCTIONS
    x = np.abs(seg(20, 33, 5)) * 3 + 0.01
    x64 = x.astype(np.float64)
    assert np.array_equal(oracle_vector_function(x, K["multiplication"], 500.0), x * np.float32(500))
    assert np.array_equal(oracle_vector_function(x * 500, K["quantize"], 1.0), np.rint(x * 500))
    assert np.array_equal(oracle_vector_function(x, K["quantize"], 0.25), (np.rint((x / np.float32(0.25)).astype(np.float64)) * 0.25).astype(np.float32))
    assert np.allclose(oracle_vector_function(x, K["log"]), np.log10(x64), rtol=1e-6, atol=1e-7)
    assert np.allclose(oracle_vector_function(x, K["power"], 0.33), x64 ** np.float64(np.float32(0.33)), rtol=1e-7)
    assert np.array_equal(oracle_vector_function(x - 1, K["maximum"], 0.5), np.maximum(x - 1, np.float32(0.5)))
    assert np.array_equal(oracle_vector_function(x - 1, K["abs"]), np.abs(x - 1))


@pytest.mark.gpu
def test_vector_functions_match_the_oracle(ctx):
    """arithmetic kinds bit-identical (incl. the f64 power and the f64 rint of quantize), the transcendental ones within a few ulp of
    glibc; strided views, in place on the identical view, overlap rejected"""
    import torch

    import rasr_amd
    from oracle.binding import oracle_vector_function
    ctx.use_torch_stream()
    K = rasr_amd.Context.VECTOR_FUNCTIONS
    n, dim = 517, 40
    x = np.abs(seg(n, dim, 91)) * 4 + 0.02
    x[5, 3] = 0.0
    wide = torch.zeros((n, dim + 3), dtype=torch.float32, device="cuda")
    wide[:, 1:1 + dim] = torch.from_numpy(x).cuda()
    for kind, prm in list(VFUNC_EXACT.items()) + list(VFUNC_ULPS.items()):
        xin = x - 1.0 if kind in ("abs", "minimum", "maximum", "addition", "quantize", "cos", "exp") else x
        wide[:, 1:1 + dim] = torch.from_numpy(xin.astype(np.float32)).cuda()
        out = torch.full((n, dim + 2), 9.0, dtype=torch.float32, device="cuda")
        ctx.vector_function(kind, prm, wide[:, 1:], dim + 3, n, dim, out[:, 1:], dim + 2)
        torch.cuda.synchronize()
        got, want = out[:, 1:1 + dim].cpu().numpy(), oracle_vector_function(xin.astype(np.float32), K[kind], prm)
        assert bool((out[:, 0] == 9.0).all()) and bool((out[:, dim + 1] == 9.0).all())
        if kind in VFUNC_EXACT:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), kind
        else:
            fin = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), fin) and np.array_equal(got[~fin], want[~fin]), kind
            assert np.all(np.abs(got[fin] - want[fin]) <= 4e-7 * np.abs(want[fin]) + 1e-7), (kind, np.abs(got - want)[fin].max())
    ctx.vector_function("multiplication", 2.0, wide[:, 1:], dim + 3, n, dim, wide[:, 1:], dim + 3)   # in place
    with pytest.raises(rasr_amd.AmxError, match="overlap"):
        ctx.vector_function("abs", 0.0, wide[:, 1:], dim + 3, n, dim, wide[:, 2:], dim + 3)
    with pytest.raises(rasr_amd.AmxError):
        ctx.vector_function("abs", 0.0, wide, dim + 3, n, dim + 4, out, dim + 2)
