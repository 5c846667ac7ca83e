"""tools/fuzz_ffnn.py [n_cases] [seed] -- random network shapes and batch sizes through the bf16 GEMM kernels: every tile configuration
(amx_ffnn_model.tuning tile = 0 / 2 / 3 / 4 / 6 and the automatic choice) must give bit-identical scores and arg-min statistics, in plain bf16 AND in
split bf16 (bf16x3) AND in f16 + MX-fp6 (f16mx); the bf16 result must stay within bf16 rounding of the fp32 MFMA path (itself checked against the oracle elsewhere),
the split-bf16 and f16mx results within 2e-4 relative of it (both are f32-accumulated; the bar against f64 accumulation is in the test suite)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from tests import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
ctx = rasr_amd.Context(0)
ctx.use_torch_stream()
bad = 0
for case in range(n_cases):
    dims = [int(rng.integers(1, 600))] + [int(rng.choice([64, 100, 257, 512, 1000, 2048])) for _ in range(int(rng.integers(0, 3)))] + \
           [int(rng.choice([1, 37, 128, 1000, 4501, 10000]))]
    T = int(rng.choice([1, 100, 256, 1024, 3000, 9000, 20000]))
    Ws, bs, acts, logp = synth.ffnn(dims, seed=int(rng.integers(1, 1000)), act=int(rng.choice([1, 2, 3])))
    x = torch.from_numpy(rng.standard_normal((T, dims[0])).astype(np.float32)).cuda()
    outs = {}
    for cfg in ("auto", "0", "2", "3", "4", "6", "fp32"):
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="fp32" if cfg == "fp32" else "bf16",
                                           tuning=None if cfg in ("auto", "fp32") else "tile=" + cfg)
        s = torch.empty((T, dims[-1]), dtype=torch.float32, device="cuda")
        best = torch.empty((T,), dtype=torch.int32, device="cuda")
        counts = torch.zeros((dims[-1],), dtype=torch.int64, device="cuda")
        ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
        nn.score_stats_dev(x, dims[0], T, s, best, counts, ssum)
        torch.cuda.synchronize()
        outs[cfg] = (s, best, counts)
        if cfg != "fp32" and not torch.equal(best.long(), s.argmin(dim=1)):
            bad += 1
            print("MISMATCH fused arg-min", cfg, dims, T)
    ref = outs["auto"]
    for cfg in ("0", "2", "3", "4", "6"):
        if not (torch.equal(outs[cfg][0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(outs[cfg][1], ref[1]) and torch.equal(outs[cfg][2], ref[2])):
            bad += 1
            print("MISMATCH config", cfg, "vs auto", dims, T, float((outs[cfg][0] - ref[0]).abs().max()))
    f32 = outs["fp32"][0]
    scale = float(f32.abs().max()) + 1.0
    err = float((ref[0] - f32).abs().max())
    if not err <= 3e-2 * scale:
        bad += 1
        print("MISMATCH bf16 vs fp32", dims, T, err, scale)
    # ---- split bf16: the same configurations, bit-identical among themselves, close to the fp32 MFMA path
    x3 = {}
    for cfg in ("auto", "0", "2", "3", "4", "6"):
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="bf16x3",
                                           tuning=None if cfg == "auto" else "tile=" + cfg)
        s = torch.empty((T, dims[-1]), dtype=torch.float32, device="cuda")
        best = torch.empty((T,), dtype=torch.int32, device="cuda")
        counts = torch.zeros((dims[-1],), dtype=torch.int64, device="cuda")
        ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
        nn.score_stats_dev(x, dims[0], T, s, best, counts, ssum)
        torch.cuda.synchronize()
        x3[cfg] = (s, best, counts)
        if not torch.equal(best.long(), s.argmin(dim=1)):
            bad += 1
            print("MISMATCH fused arg-min (bf16x3)", cfg, dims, T)
    for cfg in ("0", "2", "3", "4", "6"):
        if not (torch.equal(x3[cfg][0].view(torch.int32), x3["auto"][0].view(torch.int32)) and torch.equal(x3[cfg][1], x3["auto"][1]) and
                torch.equal(x3[cfg][2], x3["auto"][2])):
            bad += 1
            print("MISMATCH bf16x3 config", cfg, "vs auto", dims, T, float((x3[cfg][0] - x3["auto"][0]).abs().max()))
    err3 = float(((x3["auto"][0] - f32).abs() - 2e-4 * f32.abs()).max())
    if not err3 <= 2e-4:
        bad += 1
        print("MISMATCH bf16x3 vs fp32", dims, T, err3)
    # ---- f16 + MX-fp6 (AMX_PREC_F16MX): its three tile configurations bit-identical among themselves, the split-bf16 bar against fp32
    mx = {}
    for cfg in ("auto", "0", "2", "3", "8", "9"):   # 8: the 256 x 256 tile's ping-pong K loop (round 5; "auto" picks it for large batches); 9: one self-pipelined wave per SIMD
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx",
                                           tuning=None if cfg == "auto" else "tile=" + cfg)
        s = torch.empty((T, dims[-1]), dtype=torch.float32, device="cuda")
        best = torch.empty((T,), dtype=torch.int32, device="cuda")
        counts = torch.zeros((dims[-1],), dtype=torch.int64, device="cuda")
        ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
        nn.score_stats_dev(x, dims[0], T, s, best, counts, ssum)
        torch.cuda.synchronize()
        mx[cfg] = (s, best, counts)
        if not torch.equal(best.long(), s.argmin(dim=1)):
            bad += 1
            print("MISMATCH fused arg-min (f16mx)", cfg, dims, T)
    for cfg in ("0", "2", "3", "8"):
        if not (torch.equal(mx[cfg][0].view(torch.int32), mx["auto"][0].view(torch.int32)) and torch.equal(mx[cfg][1], mx["auto"][1]) and
                torch.equal(mx[cfg][2], mx["auto"][2])):
            bad += 1
            print("MISMATCH f16mx config", cfg, "vs auto", dims, T, float((mx[cfg][0] - mx["auto"][0]).abs().max()))
    errm = float(((mx["auto"][0] - f32).abs() - 2e-4 * f32.abs()).max())
    if not errm <= 2e-4:
        bad += 1
        print("MISMATCH f16mx vs fp32", dims, T, errm)
    # ---- f16mx on operands that are not Gaussian (tests/ffnn_families.py): against the exact-f32 MFMA path of the same network, errors
    # relative to the frame's score scale (1e-4 of the largest |score| of the frame); heavy-tailed weights must switch to split bf16
    from tests.ffnn_families import FAMILIES, make
    fam = FAMILIES[int(rng.integers(0, len(FAMILIES)))]
    fd = [440, int(rng.choice([256, 512, 1024])), int(rng.choice([300, 1000]))]
    fW, fb, fa, fl, fx = make(fam, fd, int(rng.choice([64, 300])), int(rng.integers(1, 10000)))
    ref32 = rasr_amd.NnBatchFeatureScorer(ctx, fW, fb, fa, log_prior=fl, precision="fp32").score(fx)
    raw = rasr_amd.NnBatchFeatureScorer(ctx, fW, fb, fa, log_prior=fl, precision="f16mx", tuning="mx_fallback=off").score(fx)
    dfl = rasr_amd.NnBatchFeatureScorer(ctx, fW, fb, fa, log_prior=fl, precision="f16mx")
    eff, ratio = dfl.effective_precision()
    scale = np.maximum(np.abs(ref32), np.abs(ref32).max(axis=1, keepdims=True))
    # (the raw scheme on heavy-tailed weights -- block ratio above 4, which the default handle does not run in f16mx -- may reach ~2.5 of
    # the frame-scale bar: profiles/r05/f16mx_families.log, log-normal rows 1.7, this fuzzer 2.2)
    for name, got, lim in (("raw f16mx", raw, 4.0 if ratio > 4.0 else 2.0), ("default (%s)" % eff, dfl.score(fx), 2.0 if eff == "f16mx" else 1.0)):
        w = float((np.abs(got - ref32) / (1e-4 * scale + 1e-4)).max())
        if not w <= lim:
            bad += 1
            print("MISMATCH family", fam, name, fd, "worst over the frame-scale bar", w)
    if (ratio > 4.0) != (eff == "bf16x3"):
        bad += 1
        print("MISMATCH family", fam, "block ratio", ratio, "but the handle runs", eff)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
