"""tools/fuzz_scorers.py [n_cases] [seed] -- random configurations of the round-2 scorers against the oracle (GPU box):
preselection-batch-float / -int (cluster counts, select counts, iterations, models with fewer densities than clusters, saturating
features), SIMD-diagonal-maximum and batch-diagonal-maximum-int on any dimension, the class-label wrapper with disregarded classes on
the NN batch scorer (fp32), the on-demand scorer on random (frame, emission) lists, the precomputed scorer on strided rows and the
decoder-side row gather.  Companion of tools/fuzz_more.py and tools/fuzz_frontends.py; exit code 1 on any mismatch."""
import os
import sys

import numpy as np
import torch  # before the library touches HIP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle import OracleGmm, nn_scorers, oracle_ffnn_score  # noqa: E402
from tests import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
ctx = rasr_amd.Context(0)
bad, ran = 0, {"presel-float": 0, "presel-int": 0, "simd": 0, "batch-int": 0, "class-labels": 0, "on-demand": 0, "precomputed": 0, "gather": 0}


def fail(what, **kw):
    global bad
    bad += 1
    print("MISMATCH", what, kw, flush=True)


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


for case in range(n_cases):
    seed = int(rng.integers(1, 1 << 30))
    # ------------------------------------------------------------------ preselection / integer scorers (pooled covariance)
    dim = int(rng.choice([8, 16, 24, 33, 39, 40, 45, 64]))
    n_mix, kmax = int(rng.integers(1, 260)), int(rng.integers(1, 20))
    model = synth.gmm_cart(n_mix, 1, kmax, dim, seed=seed, pooled=True)
    T = int(rng.choice([1, 7, 64, 65, 200, 333]))
    x = (rng.standard_normal((T, dim)) * rng.choice([0.5, 1.0, 2.0])).astype(np.float32)
    if rng.integers(0, 3) == 0:
        x[int(rng.integers(0, T))] *= 40.0   # saturates the u8 quantiser
    contract = ("off", "fma")[int(rng.integers(0, 2))]   # the reference's two arithmetics: the context's setting reaches every scorer type
    ctx.set_contract(contract)
    o = OracleGmm(model, contract=contract)
    clusters = int(rng.choice([1, 2, 8, 16, 64, 256]))
    select = int(rng.integers(1, clusters + 1))
    iters = int(rng.integers(1, 7))
    backoff = float(rng.choice([40000.0, 123.0]))
    for name, key in (("preselection-batch-float", "presel-float"), ("preselection-batch-int", "presel-int")):
        ran[key] += 1
        sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type=name)
        sc.set_preselection(clusters, select, iters, backoff)
        try:
            got = sc.score(x, want_best=False)
        except rasr_amd.AmxError as e:
            got = None
        try:
            want, wcof, wcm = (o.score_preselection_float(x, clusters, select, iters, backoff) if key == "presel-float"
                               else o.score_preselection_int(x, clusters, select, iters))
        except Exception:
            want = None
        if (got is None) != (want is None):
            fail(name + " accept/reject", product=got is not None, oracle=want is not None, n_mix=n_mix, kmax=kmax, dim=dim, clusters=clusters,
                 select=select, seed=seed)
        elif got is not None:
            cof, cm = sc.preselection_clustering()
            if not np.array_equal(cof, wcof):
                fail(name + " clustering", n_mix=n_mix, kmax=kmax, dim=dim, clusters=clusters, iters=iters, seed=seed)
            elif not same_bits(got, want):
                fail(name + " scores", n_mix=n_mix, kmax=kmax, dim=dim, clusters=clusters, select=select, iters=iters, T=T, seed=seed)
    for name, key, ref in (("SIMD-diagonal-maximum", "simd", o.score_simd), ("batch-diagonal-maximum-int", "batch-int", o.score_batch_int)):
        ran[key] += 1
        sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type=name)
        if key == "simd":
            got, gb = sc.score(x)
            want, wb, _ = ref(x)
            if not same_bits(got, want) or not np.array_equal(gb, wb):
                fail(name, n_mix=n_mix, kmax=kmax, dim=dim, T=T, seed=seed)
        else:
            got, want = sc.score(x, want_best=False), ref(x)
            if not same_bits(got, want):
                fail(name, n_mix=n_mix, kmax=kmax, dim=dim, T=T, seed=seed)
    # ------------------------------------------------------------------ NN: class labels, on-demand, precomputed, gather
    n_hidden = int(rng.integers(0, 3))
    dims = [int(rng.integers(4, 80))] + [int(rng.integers(8, 200)) for _ in range(n_hidden)] + [int(rng.integers(2, 300))]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=seed % 100000)
    n_out = dims[-1]
    n_dis = int(rng.integers(0, 4))
    n_cls = n_out + n_dis
    dis = tuple(int(v) for v in rng.choice(n_cls, n_dis, replace=False))
    mapping, nt = rasr_amd.class_labels_init(n_cls, dis)
    wmap, wnt = nn_scorers.class_labels_init(n_cls, dis)
    alpha = float(rng.choice([0.0, 0.6, 1.0]))
    Tn = int(rng.choice([1, 3, 40, 130]))
    xf = rng.standard_normal((Tn, dims[0])).astype(np.float32)
    ran["class-labels"] += 1
    if not np.array_equal(mapping, wmap) or nt != wnt:
        fail("class_labels_init", n_cls=n_cls, dis=dis)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=alpha, precision="fp32", class_to_output=mapping)
    got = nn.score(xf)
    net = oracle_ffnn_score(Ws, bs, acts, xf, log_prior=logp, prior_scale=alpha, acc64=True)
    want = nn_scorers.class_label_scores(net, mapping)
    live = mapping >= 0
    if got.shape != want.shape or not np.all(got[:, ~live] == nn_scorers.FLT_MAX) or \
            not np.all(np.abs(got[:, live] - want[:, live]) <= 1e-4 * np.abs(want[:, live]) + 1e-4):
        fail("class-label scores", dims=dims, dis=dis, alpha=alpha, T=Tn, seed=seed)
    ctx.use_torch_stream()
    ran["on-demand"] += 1
    H = nn.hidden_dim
    xd = torch.from_numpy(xf).cuda()
    act = torch.empty((Tn, H), dtype=torch.float32, device="cuda")
    nn.forward_hidden_dev(xd, dims[0], Tn, act)
    P = int(rng.integers(1, 700))
    fr, em = rng.integers(0, Tn, P).astype(np.int32), rng.integers(0, n_cls, P).astype(np.int32)
    sc = torch.empty((P,), dtype=torch.float32, device="cuda")
    nn.score_on_demand_dev(act, P, torch.from_numpy(fr).cuda(), torch.from_numpy(em).cuda(), sc)
    torch.cuda.synchronize()
    g = sc.cpu().numpy()
    folded = (bs[-1] - np.float32(alpha) * logp).astype(np.float32)
    w = nn_scorers.on_demand_scores(act.cpu().numpy(), Ws[-1], folded, fr.astype(np.uint32), em.astype(np.uint32), mapping)
    d = mapping[em] < 0
    if not np.all(g[d] == nn_scorers.FLT_MAX) or not np.all(np.abs(g[~d] - w[~d]) <= 1e-5 * np.abs(w[~d]) + 1e-5) or \
            not np.all(np.abs(g[~d] - got[fr, em][~d]) <= 1e-4 * np.abs(got[fr, em][~d]) + 1e-4):
        fail("on-demand", dims=dims, dis=dis, alpha=alpha, T=Tn, P=P, seed=seed)
    ran["precomputed"] += 1
    pad = int(rng.integers(0, 6))
    xp = (rng.standard_normal((Tn, n_out + pad)) * 7).astype(np.float32)
    out = torch.empty((Tn, n_cls), dtype=torch.float32, device="cuda")
    rasr_amd.precomputed_score_dev(ctx, torch.from_numpy(xp).cuda(), n_out + pad, Tn, n_cls, torch.from_numpy(mapping).cuda(), torch.from_numpy(logp).cuda(),
                                   alpha, out)
    torch.cuda.synchronize()
    if not same_bits(out.cpu().numpy(), nn_scorers.precomputed_scores(xp[:, :n_out], logp, alpha, mapping)):
        fail("precomputed", n_out=n_out, dis=dis, alpha=alpha, T=Tn, seed=seed)
    ran["gather"] += 1
    G = int(rng.integers(1, 400))
    rows, cols = rng.integers(0, Tn, G).astype(np.uint32), rng.integers(0, n_cls, G).astype(np.uint32)
    sub = ctx.gather_scores(torch.from_numpy(got).cuda(), n_cls, rows, cols)
    if not same_bits(np.asarray(sub, np.float32), got[rows, cols]):
        fail("gather", pairs=G, T=Tn, n_cls=n_cls)

print("fuzz_scorers: %d cases, ran %s, %d mismatches" % (n_cases, ran, bad))
sys.exit(1 if bad else 0)
