"""How close a low-precision NN path is to the reference, in the terms north_star uses (tests and bench.py's `parity_check`):
scores against the f64-accumulating oracle with and without the absolute term, arg-min state over ALL frames, and the frames a
gap rule would have to exclude.  Contract: Nn/BatchFeatureScorer.cc:92-171 (what the decoder reads)."""
import numpy as np


def nn_parity_report(got, want, other=None, gap=1e-5):
    """got, want: scores [T, n_states]; other: a second reference (the library's exact-f32 MFMA path).
    gap: frames whose two best reference scores differ by less than gap * (1 + |best|) are 'unclear' (the reference's own sgemm
    order could flip them)."""
    got = np.asarray(got)
    want = np.asarray(want, np.float64)
    err = np.abs(got.astype(np.float64) - want)
    bar = 1e-4 * np.abs(want) + 1e-4
    big = np.abs(want) > 1e-2
    rel = err[big] / np.abs(want[big])
    am_g, am_w = got.argmin(axis=1), want.argmin(axis=1)
    part = np.partition(want, 1, axis=1)[:, :2]
    rgap = (part[:, 1] - part[:, 0]) / (1.0 + np.abs(part[:, 0]))
    mism = am_g != am_w
    r = dict(frames=int(len(am_w)), scores=int(got.size),
             worst_abs=float(err.max()), worst_over_bar=float((err / bar).max()), bar_violations=int((err > bar).sum()),
             worst_pure_relative=float(rel.max()) if rel.size else 0.0, pure_relative_over_1e4=int((rel > 1e-4).sum()),
             argmin_mismatches=int(mism.sum()),
             gap_rule=gap, frames_excluded_by_gap_rule=int((rgap <= gap).sum()),
             argmin_mismatches_outside_gap_rule=int((mism & (rgap > gap)).sum()),
             largest_gap_of_a_mismatch=float(rgap[mism].max()) if mism.any() else 0.0)
    if other is not None:
        r["argmin_mismatches_vs_fp32_mfma"] = int((am_g != np.asarray(other).argmin(axis=1)).sum())
    return r
