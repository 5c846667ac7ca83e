"""tools/fuzz_more.py [n_cases] [seed] -- random configurations of the other entry points against the oracle: batch-float / batch-int /
log-add GMM scorers, Viterbi and Baum-Welch statistics, the MFCC / MF-PLP front-ends (segment lengths, sample rates, sizes), the FFNN
forward in fp32 (layer shapes, activations, priors, batch sizes that cross the tile configurations).  Companion of tools/fuzz_gmm.py."""
import os
import sys

import numpy as np
import torch  # before the library touches HIP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle import OracleGmm, OracleMfcc, oracle_ffnn_score  # noqa: E402
from oracle.binding import MfccCfg  # noqa: E402
from tests import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
ctx = rasr_amd.Context(0)
bad = 0


def fail(what, **kw):
    global bad
    bad += 1
    print("MISMATCH", what, kw)


for case in range(n_cases):
    seed = int(rng.integers(1, 1 << 30))
    # ---- pooled GMM variants
    dim = int(rng.choice([5, 8, 16, 24, 32, 33, 39, 40, 45, 48, 50, 64, 72]))
    n_mix, kmax = int(rng.integers(1, 200)), int(rng.integers(1, 30))
    model = synth.gmm_cart(n_mix, 1, kmax, dim, seed=seed, pooled=True)
    T = int(rng.choice([1, 5, 64, 129, 256, 600]))
    x = (rng.standard_normal((T, dim)) * rng.choice([0.5, 1.0, 2.0])).astype(np.float32)
    o = OracleGmm(model)
    for name, ref in (("batch-diagonal-maximum-float", o.score_batch_float), ("batch-diagonal-maximum-int", o.score_batch_int)):
        got = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type=name).score(x, want_best=False)
        if not np.array_equal(got.view(np.uint32), ref(x).view(np.uint32)):
            fail(name, dim=dim, n_mix=n_mix, kmax=kmax, T=T, seed=seed)
    got, _ = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="diagonal-sum").score(x)
    want, _ = o.score(x, mode=1)
    if not np.allclose(got, want, rtol=2e-5, atol=2e-5):
        fail("diagonal-sum", dim=dim, n_mix=n_mix, T=T, seed=seed, err=float(np.abs(got - want).max()))
    # ---- statistics
    pooled = bool(rng.integers(0, 2))
    model = synth.gmm_cart(int(rng.integers(2, 60)), 1, int(rng.integers(1, 12)), dim, seed=seed + 1, pooled=pooled)
    o = OracleGmm(model)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    nm = len(model["mix_offsets"]) - 1
    mix = rng.integers(0, nm, T).astype(np.uint32)
    w = rng.uniform(0, 2, T)
    osc, obest = o.score(x)
    chosen = obest[np.arange(T), mix].astype(np.uint32)
    ctx.use_torch_stream()
    xd, md, cd, wd = (torch.from_numpy(a).cuda() for a in (x, mix.astype(np.int32), chosen.astype(np.int32), w))
    for mode, want in ((0, o.accumulate_weighted(0, x, mix, w, chosen)), (1, o.accumulate_weighted(1, x, mix, w))):
        acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
        sc.accumulate_weighted_dev(mode, xd, T, md, wd, cd if mode == 0 else None, 0, acc)
        torch.cuda.synchronize()
        g = acc.cpu().numpy()
        tol = dict(rtol=1e-12, atol=1e-9) if mode == 0 else dict(rtol=5e-5, atol=5e-6 * max(1.0, float(np.abs(want).max())))
        if not np.allclose(g, want, **tol):
            fail("accumulate mode %d" % mode, dim=dim, pooled=pooled, T=T, seed=seed, err=float(np.abs(g - want).max()))
    # unweighted Viterbi kernel (frames of one density chained per block), both ways of passing the chosen densities
    run = np.repeat(rng.integers(0, nm, T // 5 + 1), 5)[:T].astype(np.uint32)     # bursty alignment
    chosen2 = obest[np.arange(T), run].astype(np.uint32)
    want = o.accumulate(x, run, chosen2)
    bd = torch.from_numpy(obest.astype(np.int32)).cuda()
    for ld, arg in ((nm, bd), (0, torch.from_numpy(chosen2.astype(np.int32)).cuda())):
        acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
        sc.accumulate_dev(xd, T, torch.from_numpy(run.astype(np.int32)).cuda(), arg, ld, acc)
        torch.cuda.synchronize()
        if not np.allclose(acc.cpu().numpy(), want, rtol=1e-12, atol=1e-9):
            fail("accumulate (unweighted, ld=%d)" % ld, dim=dim, pooled=pooled, T=T, seed=seed)
    # ---- front-ends
    fs = float(rng.choice([8000.0, 11025.0, 16000.0, 22050.0]))
    n = int(rng.choice([1, 159, 400, 401, 1999, 16000, 48001]))
    pcm = synth.waveform(n, seed=seed % 1000)
    if rng.integers(0, 2):
        nc = int(rng.integers(2, 17))
        cfgo = MfccCfg.default(n_ceps=nc, sample_rate=fs)
        fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=nc, sample_rate=fs)
        tol = (1e-4, 2e-3)
    else:
        nac = int(rng.integers(2, 14))
        nc = int(rng.integers(2, nac + 1))
        cfgo = MfccCfg.mfplp(n_ceps=nc, n_autocorrelation=nac, sample_rate=fs)
        fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=nc, sample_rate=fs, front_end="mfplp", nr_autocorrelation_coefficients=nac, normalize=True)
        tol = (2e-3, 2e-3)
    got, want = fe.run(pcm), OracleMfcc(cfgo).run(pcm)
    fin = np.isfinite(want)
    if got.shape != want.shape or not np.array_equal(np.isfinite(got), fin) or not np.all(np.abs(got[fin] - want[fin]) <= tol[0] * np.abs(want[fin]) + tol[1]):
        fail("front-end", fs=fs, n=n, front_end=cfgo.front_end, nc=nc, shape=(got.shape, want.shape),
             err=float(np.abs(got[fin] - want[fin]).max()) if got.shape == want.shape and fin.any() else None)
    # ---- FFNN fp32
    dims = [int(rng.integers(1, 70))] + [int(rng.integers(1, 200)) for _ in range(int(rng.integers(0, 3)))] + [int(rng.integers(1, 300))]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=seed % 997, act=int(rng.choice([1, 2, 3])))
    Tn = int(rng.choice([1, 7, 128, 129, 300, 1025]))
    xin = rng.standard_normal((Tn, dims[0])).astype(np.float32)
    ps = float(rng.choice([0.0, 0.5, 1.0]))
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=ps, precision="fp32")
    got = nn.score(xin)
    want = oracle_ffnn_score(Ws, bs, acts, xin, log_prior=logp, prior_scale=ps, acc64=True)
    if not np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4):
        fail("ffnn fp32", dims=dims, T=Tn, act=acts[0], err=float(np.abs(got - want).max()))
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
