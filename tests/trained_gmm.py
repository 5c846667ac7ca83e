"""A trained-SHAPED CART GMM built with the repository's own training loop (GPU; used by bench.py's `split-trained` config and by
tests/test_gmm_gpu.py).  The random-init model of synth.gmm_cart has 16 unrelated means per mixture -- ~1.02 of 16 densities
survive the fused scorer's screen.  A model that comes out of RASR's acoustic model trainer does not look like that: it starts
with one density per state and is grown by `split` + re-estimation (Mm/MixtureSetSplitter.cc:38-123: every mean with enough
observations becomes the twins mean +- sqrt(var) * perturbation * f32 epsilon; Mm/AbstractMixtureSetEstimator.cc:117-150 accumulate,
:305-338 estimate), so the densities of a mixture are close relatives.  This module runs that loop on synthetic clustered
features:

    1 density per state -> accumulate -> estimate      (state means, pooled covariance)
    `rounds` x [ estimate + split  ->  `iters` x ( Viterbi pass: score, best density of the aligned state, accumulate -> estimate ) ]

through GmmFeatureScorer.score_dev / accumulate_dev and rasr_amd.gmm_estimate -- the path of bench.py's gmm-train workload --
with the alignment given (frame -> state), like the reference's alignment caches.
"""
import numpy as np


def clustered_features(n_mix, dim, frames_per_state, seed, device="cuda", spread=0.7, components=16):
    """frames of state s: c_s + spread * d_{s,j} + sigma * N(0, 1), j one of `components` sub-clusters; returns (x [T, dim] f32 on
    `device`, alignment [T] int32 on `device`), frames of all states interleaved"""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    centers = torch.randn((n_mix, dim), generator=g, device=device)
    sub = torch.randn((n_mix, components, dim), generator=g, device=device) * spread
    sigma = 0.5 + torch.rand((dim,), generator=g, device=device)
    T = n_mix * frames_per_state
    align = torch.arange(T, device=device, dtype=torch.int64) % n_mix
    comp = torch.randint(0, components, (T,), generator=g, device=device)
    x = centers[align] + sub[align, comp] + sigma * torch.randn((T, dim), generator=g, device=device)
    return x.contiguous(), align.to(torch.int32).contiguous()


def initial_model(n_mix, dim):
    """one density per mixture, zero means, unit pooled variance"""
    return dict(dim=dim, mix_offsets=np.arange(n_mix + 1, dtype=np.uint32), dens_index=np.arange(n_mix, dtype=np.uint32),
                log_weight=np.zeros(n_mix, np.float64), dens_mean=np.arange(n_mix, dtype=np.uint32), dens_cov=np.zeros(n_mix, np.uint32),
                means=np.zeros((n_mix, dim), np.float32), variances=np.ones((1, dim), np.float32))


def viterbi_pass(ctx, model, x, align, chunk=32768):
    """one pass over the data: every frame's best density of its aligned mixture (diagonal-maximum) -> statistics; returns the flat
    f64 accumulator (host) and the mean score of the aligned mixtures"""
    import torch

    import rasr_amd
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    M = sc.nMixtures()
    T = x.shape[0]
    acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
    n = min(chunk, T)
    scores = torch.empty((n, M), dtype=torch.float32, device="cuda")
    bestd = torch.empty((n, M), dtype=torch.int32, device="cuda")
    total = 0.0
    for t0 in range(0, T, chunk):
        m = min(chunk, T - t0)
        sc.score_dev(x[t0:], m, scores, bestd)
        sc.accumulate_dev(x[t0:], m, align[t0:], bestd, M, acc)
        total += float(scores[:m].gather(1, align[t0:t0 + m].long()[:, None]).double().sum())
    torch.cuda.synchronize()
    return acc.cpu().numpy(), total / T


def split_trained_gmm(ctx, n_mix=10000, dim=40, frames_per_state=600, rounds=4, iters=2, seed=11, log=None):
    """-> (model dict with up to 2^rounds densities per mixture and a pooled covariance, x, align, history); the model right after
    the last split (exact twins, before any re-estimation) is left in split_trained_gmm.fresh_split"""
    import rasr_amd
    ctx.use_torch_stream()
    x, align = clustered_features(n_mix, dim, frames_per_state, seed)
    model = initial_model(n_mix, dim)
    hist = []
    split_trained_gmm.fresh_split = None
    acc, s = viterbi_pass(ctx, model, x, align)
    for r in range(rounds):
        model = rasr_amd.gmm_estimate(model, acc, split=1)
        if r == rounds - 1:   # the model as the trainer writes it right after its last split: twins mean +- eps, not yet re-estimated
            split_trained_gmm.fresh_split = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in model.items()}
        for it in range(iters):
            acc, s = viterbi_pass(ctx, model, x, align)
            if it < iters - 1:   # the last pass's statistics go into the next round's estimate + split (or the final estimate)
                model = rasr_amd.gmm_estimate(model, acc)
        hist.append(dict(round=r + 1, densities=int(model["mix_offsets"][-1]), mean_score=round(s, 4)))
        if log:
            log("split round %d: %d densities, mean score of the aligned state %.4f" % (r + 1, hist[-1]["densities"], s))
    model = rasr_amd.gmm_estimate(model, acc)   # the statistics of the last pass
    return model, x, align, hist
