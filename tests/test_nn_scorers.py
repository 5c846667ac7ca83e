"""Nn::ClassLabelWrapper, Nn::Prior file IO, on-demand and precomputed NN feature scorers."""
import numpy as np
import pytest

from tests import synth


def feats(T, dim, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((T, dim)).astype(np.float32)


# ---------------------------------------------------------------- host code (no GPU)

def test_class_labels_init_matches_the_restatement():
    import rasr_amd
    from oracle import nn_scorers
    for n, dis in [(10, ()), (10, (0,)), (53, (3, 17, 40)), (5, (0, 1, 2, 3, 4)), (7, (6, 6, 2))]:
        got, nt = rasr_amd.class_labels_init(n, dis)
        want, wnt = nn_scorers.class_labels_init(n, dis)
        assert np.array_equal(got, want) and nt == wnt


def test_vector_files_round_trip_and_reference_documents(tmp_path):
    """xml (default) and bin: forms of Math::Vector<f32> / <s32>; a document as Core::XmlWriter writes it; the size check"""
    import rasr_amd
    from oracle import nn_scorers
    rng = np.random.Generator(np.random.PCG64(5))
    pr = np.log(rng.dirichlet(np.ones(37))).astype(np.float32)
    for name in ("prior.xml", "bin:" + str(tmp_path / "prior.bin")):
        path = name if name.startswith("bin:") else str(tmp_path / name)
        rasr_amd.write_prior(path, pr)
        back = rasr_amd.read_prior(path)
        assert np.array_equal(back.view(np.uint32), pr.view(np.uint32))
    raw = open(tmp_path / "prior.bin", "rb").read()
    assert raw[:4] == (37).to_bytes(4, "little") and raw[4:] == pr.tobytes()     # Math::Vector::write: u32 size, elements
    mapping = np.array([0, -1, 1, 2, -1, 3], np.int32)
    rasr_amd.write_class_labels(str(tmp_path / "labels.xml"), mapping)
    assert np.array_equal(rasr_amd.read_class_labels(str(tmp_path / "labels.xml")), mapping)
    # documents in the reference writer's form (6 significant digits, size attribute)
    (tmp_path / "ref.xml").write_text(nn_scorers.vector_xml(pr, "f32"))
    got = rasr_amd.read_prior(str(tmp_path / "ref.xml"))
    assert np.allclose(got, pr, rtol=1e-6)
    (tmp_path / "ref_s32.xml").write_text(nn_scorers.vector_xml(mapping, "s32"))
    assert np.array_equal(rasr_amd.read_class_labels("xml:" + str(tmp_path / "ref_s32.xml")), mapping)
    (tmp_path / "bad.xml").write_text('<?xml version="1.0"?>\n<vector-f32 size="4"> 1 2 3 </vector-f32>\n')
    with pytest.raises(rasr_amd.AmxError, match="Vector dimension mismatch"):
        rasr_amd.read_prior(str(tmp_path / "bad.xml"))
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.read_prior(str(tmp_path / "missing.xml"))
    (tmp_path / "other.xml").write_text('<?xml version="1.0"?>\n<matrix-f32 nRows="1" nColumns="1"> 1 </matrix-f32>\n')
    with pytest.raises(rasr_amd.AmxError, match="vector-f32"):
        rasr_amd.read_prior(str(tmp_path / "other.xml"))


# ---------------------------------------------------------------- device paths

@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16x3", 1e-4), ("bf16", None)])
def test_class_label_mapping_and_disregarded_classes(ctx, precision, tol):
    """53 classes, 3 of them disregarded, over a 50-output network: the scores of a class are the scores of its output, a
    disregarded class scores FLT_MAX exactly, the arg-min statistics run over emissions"""
    import torch

    import rasr_amd
    from oracle import nn_scorers, oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([40, 96, 50], seed=31)
    mapping, nt = rasr_amd.class_labels_init(53, (3, 17, 40))
    assert nt == 50
    x = feats(300, 40, 32)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=0.7, precision=precision, class_to_output=mapping)
    assert nn.nMixtures() == 53
    got = nn.score(x)
    plain = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=0.7, precision=precision).score(x)
    assert np.array_equal(got.view(np.uint32), nn_scorers.class_label_scores(plain, mapping).view(np.uint32))
    assert np.all(got[:, [3, 17, 40]] == nn_scorers.FLT_MAX)
    if tol is not None:
        want = nn_scorers.class_label_scores(oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=0.7, acc64=True), mapping)
        keep = mapping >= 0
        assert np.all(np.abs(got[:, keep] - want[:, keep]) <= tol * np.abs(want[:, keep]) + tol)
    xd = torch.from_numpy(x).cuda()
    scores = torch.empty((300, 53), dtype=torch.float32, device="cuda")
    state = torch.empty((300,), dtype=torch.int32, device="cuda")
    counts = torch.zeros((53,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    nn.score_stats_dev(xd, 40, 300, scores, state, counts, ssum)
    torch.cuda.synchronize()
    assert np.array_equal(state.cpu().numpy(), got.argmin(axis=1))
    assert counts.cpu().numpy()[[3, 17, 40]].sum() == 0


@pytest.mark.gpu
def test_class_label_errors(ctx):
    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([8, 16, 5], seed=33)
    with pytest.raises(rasr_amd.AmxError, match="one-to-one"):
        rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, precision="fp32", class_to_output=np.array([0, 1, 1, 2, 3, 4], np.int32))
    with pytest.raises(rasr_amd.AmxError, match="classes to accumulate"):
        rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, precision="fp32", class_to_output=np.array([0, 1, -1, 2, 3], np.int32))
    with pytest.raises(rasr_amd.AmxError, match="maps to output"):
        rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, precision="fp32", class_to_output=np.array([0, 1, 2, 3, 9], np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("dims", [[40, 96, 64, 50], [24, 30]])
def test_on_demand_scorer(ctx, precision, dims):
    """hidden layers once per frame, output layer for requested (frame, emission) pairs only: equal to the restatement of
    LinearAndSoftmaxLayer::getScore on the exported activations (1e-5), equal to the batch scorer to its precision, FLT_MAX for
    disregarded classes; a network without hidden layers feeds the features straight into the output layer"""
    import torch

    import rasr_amd
    from oracle import nn_scorers, oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn(dims, seed=41)
    n_out = dims[-1]
    mapping, _ = rasr_amd.class_labels_init(n_out + 2, (1, n_out))
    T = 77
    x = feats(T, dims[0], 42)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=0.9, precision=precision, class_to_output=mapping)
    H = nn.hidden_dim
    assert H == dims[-2]
    xd = torch.from_numpy(x).cuda()
    act = torch.empty((T, H), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    nn.forward_hidden_dev(xd, dims[0], T, act)
    rng = np.random.Generator(np.random.PCG64(43))
    P = 500
    fr = rng.integers(0, T, P).astype(np.uint32)
    em = rng.integers(0, n_out + 2, P).astype(np.uint32)
    em[:4] = [1, n_out, 0, n_out + 1]
    sc = torch.empty((P,), dtype=torch.float32, device="cuda")
    nn.score_on_demand_dev(act, P, torch.from_numpy(fr.astype(np.int32)).cuda(), torch.from_numpy(em.astype(np.int32)).cuda(), sc)
    torch.cuda.synchronize()
    got = sc.cpu().numpy()
    a = act.cpu().numpy()
    folded = (bs[-1] - np.float32(0.9) * logp).astype(np.float32)
    want = nn_scorers.on_demand_scores(a, Ws[-1], folded, fr, em, mapping)
    dis = mapping[em] < 0
    assert np.all(got[dis] == nn_scorers.FLT_MAX) and dis.sum() >= 2
    assert np.all(np.abs(got[~dis] - want[~dis]) <= 1e-5 * np.abs(want[~dis]) + 1e-5)
    full = nn.score(x)[fr, em]
    tol = 1e-4 if precision != "bf16" else 5e-2
    assert np.all(np.abs(got[~dis] - full[~dis]) <= tol * np.abs(full[~dis]) + tol)
    if precision != "bf16" and len(dims) > 2:   # hidden activations against the f64-accumulating oracle of the truncated network
        hid = -oracle_ffnn_score(Ws[:-1], bs[:-1], acts[:-2] + [0], x, acc64=True)
        hid = np.maximum(hid, 0) if acts[-2] == 1 else hid
        assert np.all(np.abs(a - hid) <= 1e-4 * np.abs(hid) + 1e-4)


@pytest.mark.gpu
def test_precomputed_scorer_bit_exact(ctx):
    """-x[out(e)] + alpha * logPrior[out(e)] in f32 (two roundings), FLT_MAX for disregarded classes; feature rows with a stride"""
    import torch

    import rasr_amd
    from oracle import nn_scorers
    rng = np.random.Generator(np.random.PCG64(51))
    T, n_out = 130, 47
    x = rng.standard_normal((T, n_out + 5)).astype(np.float32) * 7
    logp = np.log(rng.dirichlet(np.ones(n_out))).astype(np.float32)
    mapping, _ = rasr_amd.class_labels_init(n_out + 3, (0, 20, 49))
    ctx.use_torch_stream()
    xd = torch.from_numpy(x).cuda()
    lp = torch.from_numpy(logp).cuda()
    for mp in (mapping, None):
        ncl = n_out if mp is None else len(mp)
        md = None if mp is None else torch.from_numpy(mp).cuda()
        out = torch.empty((T, ncl), dtype=torch.float32, device="cuda")
        rasr_amd.precomputed_score_dev(ctx, xd, n_out + 5, T, ncl, md, lp, 0.6, out)
        torch.cuda.synchronize()
        want = nn_scorers.precomputed_scores(x[:, :n_out], logp, 0.6, mp)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------- neural-network-forward (the Flow node in front of the precomputed scorer)

def test_oracle_softmax_rows_against_the_definition():
    from oracle.binding import oracle_ffnn_forward, oracle_softmax_rows
    rng = np.random.Generator(np.random.PCG64(61))
    x = (rng.standard_normal((37, 301)) * 5).astype(np.float32)
    y = oracle_softmax_rows(x)
    e = np.exp(x.astype(np.float64) - x.max(1, keepdims=True))
    assert np.allclose(y, e / e.sum(1, keepdims=True), rtol=2e-6, atol=1e-12)
    assert np.allclose(y.sum(1), 1, atol=1e-5) and np.array_equal(y.argmax(1), x.argmax(1))
    Ws, bs, acts, logp = synth.ffnn([12, 20, 9], seed=62)
    f = feats(5, 12, 63)
    from oracle import oracle_ffnn_score
    lin = oracle_ffnn_forward(Ws, bs, acts, f, 0, log_prior=logp, prior_scale=0.7)
    assert np.array_equal(lin, -oracle_ffnn_score(Ws, bs, acts, f, log_prior=logp, prior_scale=0.7))
    assert np.array_equal(oracle_ffnn_forward(Ws, bs, acts, f, 1, log_prior=logp, prior_scale=0.7).view(np.uint32), oracle_softmax_rows(lin).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 1e-5), ("bf16x3", 1e-4), ("bf16", 5e-2)])
def test_forward_node_outputs(ctx, precision, tol):
    """amx_ffnn_forward_dev: the linear top layer is exactly the negated score; the softmax of THOSE activations is bit-identical to the
    oracle's restatement of FastMatrix::softmax (maximum, f64 exponential narrowed, sequential f32 sum, reciprocal multiply); against the
    oracle's own forward pass the outputs agree to the precision of the GEMM path; a handle with a class mapping is refused"""
    import torch

    import rasr_amd
    from oracle.binding import oracle_ffnn_forward, oracle_softmax_rows
    dims = [40, 96, 64, 257]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=71)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=0.8, precision=precision)
    ctx.use_torch_stream()
    for T in (1, 77, 300):
        x = feats(T, 40, 72 + T)
        xd = torch.from_numpy(x).cuda()
        sc = torch.empty((T, 257), dtype=torch.float32, device="cuda")
        lin = torch.empty((T, 257), dtype=torch.float32, device="cuda")
        sm = torch.empty((T, 257), dtype=torch.float32, device="cuda")
        nn.score_dev(xd, 40, T, sc)
        nn.forward_dev(xd, 40, T, lin, top="linear")
        nn.forward_dev(xd, 40, T, sm, top="softmax")
        torch.cuda.synchronize()
        a = lin.cpu().numpy()
        assert np.array_equal(a.view(np.uint32), (-sc.cpu().numpy()).view(np.uint32))
        assert np.array_equal(sm.cpu().numpy().view(np.uint32), oracle_softmax_rows(a).view(np.uint32))
        want = oracle_ffnn_forward(Ws, bs, acts, x, 1, log_prior=logp, prior_scale=0.8, acc64=True)
        assert np.all(np.abs(sm.cpu().numpy() - want) <= tol * np.abs(want) + tol * 1e-2)
    mapping, _ = rasr_amd.class_labels_init(258, (3,))
    mapped = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision=precision, class_to_output=mapping)
    with pytest.raises(rasr_amd.AmxError, match="class-label mapping"):
        mapped.forward_dev(xd, 40, T, torch.empty((T, 258), dtype=torch.float32, device="cuda"))
