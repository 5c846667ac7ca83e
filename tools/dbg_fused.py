import numpy as np, sys
sys.path.insert(0, '.')
import rasr_amd
from tests import synth
from oracle import OracleGmm
ctx = rasr_amd.Context(0)
for n_mix, T in [(16, 32), (48, 256), (333, 700)]:
    model = synth.gmm_cart(n_mix, 16, 16, 40, seed=400 + n_mix, pooled=True)
    x = np.random.Generator(np.random.PCG64(1)).standard_normal((T, 40)).astype(np.float32)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    sc.screen_counts(True)
    s, b = sc.score(x)
    surv, pairs = sc.screen_counts(True)
    osc, ob = OracleGmm(model).score(x, mode=0)
    bad = (s.view(np.uint32) != osc.view(np.uint32))
    print(n_mix, T, "survivors/pair", surv / max(pairs, 1), "score mismatches", bad.sum(), "of", bad.size, "best mismatches", (b != ob).sum(), flush=True)
    if bad.any():
        i = np.argwhere(bad)[:5]
        for t, m in i:
            print("  t", t, "m", m, s[t, m], osc[t, m], b[t, m], ob[t, m])
