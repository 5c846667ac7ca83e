#!/usr/bin/env python3
"""CPU emulation of the round-4 candidate arithmetic for the NN leg, before any kernel is written:

    x . w  ~=  hi16(x) . hi16(w)                       one f16 MFMA product      (v_mfma_f32_32x32x16_f16, 1 unit)
             + q_hi(x) . q_lo(w) + q_lo(x) . q_hi(w)   two low-precision products (v_mfma_scale_f32_32x32x64_f8f6f4, 1/2 or 1/4 unit each)

with hi16 = round-to-nearest f16, lo = x - hi16 (exact in f32) and q_* the 8 / 6 / 4-bit formats of the MX instruction, scaled per
tensor (weights: one power of two per layer) or per 32-element block along K (activations: OCP MX shared exponents, what the GEMM
epilogue can compute locally).  The products of the rounded operands are exact in f32; sums run in f32 (numpy sgemm) like the MFMA
accumulators.  Everything else -- bias, ReLU, prior, negation -- is f32 as in the kernel epilogues.

Workload: BASELINE config 4 at full size (440-6x2048-10000, batch 1024, the seeds of tests/test_ffnn_gpu.py), all 10.24 M scores
against the f64-accumulating oracle (oracle_ffnn_score(acc64=True)), i.e. exactly what test_*_config4_full_size_against_the_oracle
compares.  Reference arithmetic replaced: Nn/LinearLayer.cc:298-324, Math/Blas.hh:402-420.

    python tools/emulate_split_f16_f8.py [--out profiles/r04/emulation_f16_f8.json]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tests import synth  # noqa: E402

FORMATS = {  # name: (mantissa bits, emin (unbiased exponent of the smallest normal), largest finite value, its exponent)
    "e4m3": (3, -6, 448.0, 8),
    "e5m2": (2, -14, 57344.0, 15),
    "e2m3": (3, 0, 7.5, 2),     # fp6
    "e3m2": (2, -2, 28.0, 4),   # bf6
    "e2m1": (1, 0, 6.0, 2),     # fp4
}


def minifloat(v, fmt, trunc=False):
    """round v (f64) to the format: nearest-even (or toward zero), gradual underflow, saturating"""
    mb, emin, vmax, _ = FORMATS[fmt]
    a = np.abs(v)
    e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, emin)
    step = np.exp2(e - mb)
    q = (np.floor(a / step) if trunc else np.rint(a / step)) * step
    return np.sign(v) * np.minimum(q, vmax)


def pair_max(m):
    """block maxima [rows, blocks, 1] -> the larger of rows r and r ^ 32 (the two 32-row blocks one lane of the GEMM serves): ONE
    conversion instruction converts both rows' 16 values and takes one scale"""
    r = m.shape[0]
    pad = (-r) % 64
    mp = np.pad(m, ((0, pad), (0, 0), (0, 0))) if pad else m
    g = mp.reshape(-1, 2, 32, m.shape[1], 1)
    g = np.maximum(g[:, 0], g[:, 1])
    return np.repeat(g[:, None], 2, axis=1).reshape(mp.shape)[:r]


def quant(v, fmt, scaling, trunc=False, tied_to=None, tie_offset=0, pair=False):
    """quantise v [rows, K] (f32) to fmt; scaling: 'tensor' = one power of two for the whole array (max -> top binade),
    'block' = OCP MX: per row and 32-element K block the shared exponent floor(log2 max) - emax(fmt).  Returns the dequantised
    values (f32: exactly what the scaled MFMA multiplies)."""
    _, _, _, emax = FORMATS[fmt]
    v64 = v.astype(np.float64)
    if scaling == "tensor":
        m = np.abs(v64).max()
        s = np.exp2(np.floor(np.log2(m)) - emax) if m > 0 else 1.0
        return (minifloat(v64 / s, fmt, trunc) * s).astype(np.float32)
    r, k = v.shape
    pad = (-k) % 32
    if pad:
        v64 = np.pad(v64, ((0, 0), (0, pad)))
    b = v64.reshape(r, -1, 32)
    if tied_to is not None:  # the block exponent of ANOTHER array (the values whose residual v is), shifted: what the GEMM epilogue does
        t64 = tied_to.astype(np.float64)
        if pad:
            t64 = np.pad(t64, ((0, 0), (0, pad)))
        m = np.abs(t64.reshape(r, -1, 32)).max(axis=2, keepdims=True) * np.exp2(tie_offset)
    else:
        m = np.abs(b).max(axis=2, keepdims=True)
    if pair:
        m = pair_max(m)
    s = np.exp2(np.floor(np.log2(np.where(m > 0, m, 1.0))) - emax)
    q = (minifloat(b / s, fmt, trunc) * s).reshape(r, -1)[:, :k]
    return q.astype(np.float32)


def e5m2_of_f16_top_byte(h16):
    """the top byte of the f16 encoding read as e5m2 (truncation toward zero): what a v_perm_b32 of hi16 fragments yields"""
    bits = h16.view(np.uint16) & np.uint16(0xFF00)
    return bits.view(np.float16).astype(np.float32)


def split(v, hi_fmt, lo_fmt, scaling, tie_lo=False, pair=False):
    """-> hi16 (f32 values), q_hi, q_lo"""
    h16 = v.astype(np.float16)
    hi = h16.astype(np.float32)
    lo = v - hi                                   # exact
    if hi_fmt == "e5m2-top-byte":
        qh = e5m2_of_f16_top_byte(h16)
    elif hi_fmt is None:
        qh = None
    else:
        qh = quant(hi, hi_fmt, scaling, tied_to=v if pair else None, pair=pair)
    if lo_fmt is None:
        ql = None
    elif tie_lo:  # lo's block scale = hi's block scale 2^-11 (|lo| <= 2^-11 2^E: never saturates), no second maximum
        ql = quant(lo, lo_fmt, "block", tied_to=v, tie_offset=-11, pair=pair)
    else:
        ql = quant(lo, lo_fmt, scaling)
    return hi, qh, ql


def bf16_round(v):
    u = v.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def forward(Ws, bs, acts, logp, x, scheme):
    """scheme: dict(kind='f16x' | 'bf16x3' | 'f16', hi=fmt, lo=fmt, x_scaling=..., w_scaling=...)"""
    y = x
    n = len(Ws)
    for l in range(n):
        W = Ws[l]
        if scheme["kind"] == "bf16x3":
            wh = bf16_round(W); wl = bf16_round(W - wh)
            xh = bf16_round(y); xl = bf16_round(y - xh)
            z = xh @ wh.T + (xh @ wl.T + xl @ wh.T)
        else:
            wh, wqh, wql = split(W, scheme.get("hi"), scheme.get("lo"), scheme.get("w_scaling", "tensor"), scheme.get("tie_w", False), scheme.get("pair", False) or scheme.get("pair_w", False))
            xh, xqh, xql = split(y, scheme.get("hi"), scheme.get("lo"), scheme.get("x_scaling", "block"), scheme.get("tie_lo", False), scheme.get("pair", False))
            z = xh @ wh.T
            if scheme["kind"] == "f16x":
                cross = scheme.get("cross", "both")   # round 6: "w" = q(x) r(w) only (the weights' residual), "x" = r(x) q(w) only
                if cross == "both":
                    z = z + (xqh @ wql.T + xql @ wqh.T)
                elif cross == "w":
                    z = z + xqh @ wql.T
                else:
                    z = z + xql @ wqh.T
        z = z + bs[l][None, :]
        if l < n - 1:
            y = np.maximum(z, 0.0).astype(np.float32) if acts[l] == 1 else z
        else:
            return (-(z - logp[None, :])).astype(np.float32)


def report(name, got, want, fp32=None):
    err = np.abs(got.astype(np.float64) - want)
    bar = 1e-4 * np.abs(want) + 1e-4
    big = np.abs(want) > 1e-2
    rel = err[big] / np.abs(want[big])
    am_g, am_w = got.argmin(axis=1), want.argmin(axis=1)
    srt = np.sort(want, axis=1)
    gap = (srt[:, 1] - srt[:, 0]) / (1 + np.abs(srt[:, 0]))
    mism = am_g != am_w
    edges = [0, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e9]
    hist = np.histogram(rel, bins=edges)[0]
    r = dict(scheme=name, n_scores=int(got.size), max_abs=float(err.max()), rms_abs=float(np.sqrt((err ** 2).mean())),
             worst_over_bar=float((err / bar).max()), violations_of_1e4_bar=int((err > bar).sum()),
             worst_pure_relative=float(rel.max()), rms_pure_relative=float(np.sqrt((rel ** 2).mean())),
             pure_relative_over_1e4=int((rel > 1e-4).sum()),
             argmin_mismatches_of_all_frames=int(mism.sum()), n_frames=int(len(am_w)),
             largest_gap_among_mismatches=float(gap[mism].max()) if mism.any() else 0.0,
             relative_error_histogram=dict(edges=edges, counts=hist.tolist()))
    if fp32 is not None:
        r["argmin_mismatches_vs_fp32_sgemm"] = int((am_g != fp32.argmin(axis=1)).sum())
    print(json.dumps({k: v for k, v in r.items() if k != "relative_error_histogram"}))
    return r


ONE_SIDED = [   # (name, nominal matrix units per f32 product, scheme)
    ("f16 alone", 1.0, dict(kind="f16")),
    ("f16 + q(x) r(w): the WEIGHTS' residual only (one 64-deep fp6 product per 64 k)", 1.25,
     dict(kind="f16x", hi="e2m3", lo="e2m3", w_scaling="block", tie_lo=True, tie_w=True, pair_w=True, cross="w")),
    ("f16 + r(x) q(w): the ACTIVATIONS' residual only", 1.25,
     dict(kind="f16x", hi="e2m3", lo="e2m3", w_scaling="block", tie_lo=True, tie_w=True, pair_w=True, cross="x")),
    ("f16 + both cross terms (AMX_PREC_F16MX as built)", 1.5,
     dict(kind="f16x", hi="e2m3", lo="e2m3", w_scaling="block", tie_lo=True, tie_w=True, pair_w=True)),
    ("bf16x3", 3.0, dict(kind="bf16x3")),
]


def families(a):
    """round-5 review, item 6: is there a scheme under 1.5 nominal units that meets 1e-4 PURE relative (over |ref| > 1e-2) and the
    1e-4 |ref| + 1e-4 bar on every operand family?  One cross term instead of two halves the scaled products (a 32x32x64 fp6 product
    then covers 64 k of ONE term): 1.25 units.  The gate is evaluated here, before any kernel."""
    from oracle import oracle_ffnn_score
    from tests.ffnn_families import FAMILIES, make
    table = {}
    cases = [(f, make(f, [440, 768, 768, 1500], 384, 300 + len(f))) for f in FAMILIES]
    dims = [440] + [2048] * 6 + [10000]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
    x = np.random.Generator(np.random.PCG64(6)).standard_normal((256, 440)).astype(np.float32)
    cases.append(("config 4 (440-6x2048-10000, 256 frames)", (Ws, bs, acts, logp, x)))
    for fam, (Ws, bs, acts, logp, x) in cases:
        want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True).astype(np.float64)
        f32 = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=False)
        e32 = np.abs(f32.astype(np.float64) - want)
        big = np.abs(want) > 1e-2
        row = {"f32 accumulation (the reference's own arithmetic)": dict(units=None, worst_over_bar=float((e32 / (1e-4 * np.abs(want) + 1e-4)).max()),
                                                                         worst_pure_relative=float((e32[big] / np.abs(want[big])).max()) if big.any() else 0.0,
                                                                         argmin_mismatches=int((f32.argmin(axis=1) != want.argmin(axis=1)).sum()))}
        for name, units, sch in ONE_SIDED:
            got = forward(Ws, bs, acts, logp, x, sch)
            err = np.abs(got.astype(np.float64) - want)
            big = np.abs(want) > 1e-2
            row[name] = dict(units=units, worst_over_bar=float((err / (1e-4 * np.abs(want) + 1e-4)).max()),
                             worst_pure_relative=float((err[big] / np.abs(want[big])).max()) if big.any() else 0.0,
                             argmin_mismatches=int((got.argmin(axis=1) != want.argmin(axis=1)).sum()))
        table[fam] = row
        print(fam)
        for k, v in row.items():
            print("    %-86s units %-5s worst/bar %8.3g  pure rel %8.3g" % (k[:86], v["units"], v["worst_over_bar"], v.get("worst_pure_relative", float("nan"))))
    # the gate is asked where it CAN be met: on the families whose scores f32 accumulation itself -- the reference's sgemm -- keeps inside
    # both bars (the others are ill-conditioned: a score there is a difference of terms 10^3 .. 10^5 times its size)
    F32 = "f32 accumulation (the reference's own arithmetic)"
    fair = [f for f in table if table[f][F32]["worst_over_bar"] <= 1.0 and table[f][F32]["worst_pure_relative"] <= 1e-4]
    gate = {"families_where_f32_accumulation_meets_both_bars": fair}
    for name, units, _ in ONE_SIDED:
        fams_ok = [f for f in fair if table[f][name]["worst_over_bar"] <= 1.0 and table[f][name]["worst_pure_relative"] <= 1e-4]
        gate[name] = dict(units=units, fair_families_meeting_both_bars=len(fams_ok), of=len(fair), failing=[f for f in fair if f not in fams_ok],
                          worst_over_bar_on_fair_families=max(table[f][name]["worst_over_bar"] for f in fair))
    print(json.dumps(gate, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(dict(question="a precision mode under 1.5 nominal matrix units per product that meets 1e-4 pure relative and the 1e-4 |ref| + 1e-4 bar on the ten "
                                "operand families (round-5 review, next #6)", gate=gate, table=table), open(a.out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--only", default=None, help="substring filter on the scheme names")
    ap.add_argument("--families", action="store_true", help="round 6: the ONE-SIDED corrections (1.25 nominal units per product) against the built scheme "
                                                            "on the ten operand families of tests/ffnn_families.py and on config 4")
    a = ap.parse_args()
    if a.families:
        return families(a)
    from oracle import oracle_ffnn_score
    dims = [440] + [2048] * 6 + [10000]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
    x = np.random.Generator(np.random.PCG64(6)).standard_normal((a.frames, 440)).astype(np.float32)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True).astype(np.float64)
    fp32 = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=False)
    schemes = [
        ("fp32 sgemm (reference arithmetic, f32 accumulation)", None),
        ("bf16x3 (round 3: three bf16 products)", dict(kind="bf16x3")),
        ("f16 alone (one product)", dict(kind="f16")),
        ("f16 + e4m3 x e4m3 cross terms, MX blocks on activations", dict(kind="f16x", hi="e4m3", lo="e4m3")),
        ("f16 + e4m3 x e4m3 cross terms, per-tensor scales both", dict(kind="f16x", hi="e4m3", lo="e4m3", x_scaling="tensor")),
        ("f16 + e5m2(top byte of hi16, truncated) x e4m3 lo", dict(kind="f16x", hi="e5m2-top-byte", lo="e4m3")),
        ("f16 + e5m2 (rounded) x e4m3 lo", dict(kind="f16x", hi="e5m2", lo="e4m3")),
        ("f16 + fp6 e2m3 x e2m3 cross terms, MX blocks both", dict(kind="f16x", hi="e2m3", lo="e2m3", w_scaling="block")),
        ("f16 + fp6 e3m2 x e3m2 cross terms, MX blocks both", dict(kind="f16x", hi="e3m2", lo="e3m2", w_scaling="block")),
        ("f16 + e4m3 hi x fp4 e2m1 lo, MX blocks both", dict(kind="f16x", hi="e4m3", lo="e2m1", w_scaling="block")),
        ("f16 + fp4 e2m1 x e2m1 cross terms, MX blocks both", dict(kind="f16x", hi="e2m1", lo="e2m1", w_scaling="block")),
        ("f16 + fp4 e2m1 x e2m1, MX blocks both, activation lo scale tied to the hi scale (2^-11): the scheme built as AMX_PREC_F16MX4",
         dict(kind="f16x", hi="e2m1", lo="e2m1", w_scaling="block", tie_lo=True)),
        ("f16 + fp6 e2m3 x e2m3, block scales, residual scale tied to the block scale (2^-11) on both operands, an exponent per row and block",
         dict(kind="f16x", hi="e2m3", lo="e2m3", w_scaling="block", tie_lo=True, tie_w=True)),
        ("the same with ONE block exponent per pair of WEIGHT rows n, n ^ 32 (one conversion instruction serves both: 4 instead of 6 per K-tile): AMX_PREC_F16MX as built",
         dict(kind="f16x", hi="e2m3", lo="e2m3", w_scaling="block", tie_lo=True, tie_w=True, pair_w=True)),
        ("the same with frames t, t ^ 32 paired as well (3 conversions; not built: a frame's scores would depend on its batch)",
         dict(kind="f16x", hi="e2m3", lo="e2m3", w_scaling="block", tie_lo=True, tie_w=True, pair=True)),
    ]
    out = []
    for name, sch in schemes:
        if a.only and a.only not in name and sch is not None:
            continue
        got = fp32 if sch is None else forward(Ws, bs, acts, logp, x, sch)
        out.append(report(name, got, want, fp32))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(dict(workload="BASELINE config 4: FFNN 440-6x2048-10000, batch %d, synth.ffnn(seed 7), features PCG64(6)" % a.frames,
                       reference="oracle_ffnn_score(acc64=True)", bar="|d| <= 1e-4 |ref| + 1e-4; pure relative over |ref| > 1e-2",
                       results=out), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
