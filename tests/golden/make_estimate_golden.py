"""Regenerates tests/golden/ref_estimate.json.  Run in the build container only (needs oracle/_ref/libref.so):

    python tests/golden/make_estimate_golden.py

The expected values are OUTPUTS OF THE REFERENCE ITSELF: the functors and helpers of the header-only Mm/Utilities.hh
(unrolledTransform with std::plus / plusWeighted / plusSquare / plusSquareWeighted / plusNormalizedSquare,
normalizedMinus, logExpNorm) compiled unmodified into libref.so (oracle/ref/ref_harness.cc: ref_accumulate_vector,
ref_normalized_minus, ref_log_exp_norm).  Only inputs and the expected outputs are stored, as hex of the IEEE bytes.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.binding import load_ref, ref_accumulate_vector, ref_log_exp_norm, ref_normalized_minus  # noqa: E402


def hx(a, t):
    return np.ascontiguousarray(a, dtype=t).tobytes().hex()


def main():
    if load_ref() is None:
        raise SystemExit("oracle/_ref/libref.so not available (needs /root/reference)")
    rng = np.random.Generator(np.random.PCG64(91))
    acc = []
    for dim in (1, 3, 8, 9, 40, 45):                     # unrolledTransform's Duff device: every remainder class
        for kind in (0, 1, 2, 3, 4):
            s = rng.standard_normal(dim) * 100
            v = rng.standard_normal(dim) * 3 if kind == 4 else (rng.standard_normal(dim) * 3).astype(np.float32)
            w = float(rng.uniform(0.01, 2.5))
            acc.append({"dim": dim, "kind": kind, "weight": hx([w], "<f8"), "sum": hx(s, "<f8"),
                        "v": hx(v, "<f8" if kind == 4 else "<f4"), "out": hx(ref_accumulate_vector(s, v, w, kind), "<f8")})
    norms = []
    for n in (1, 2, 5, 16, 64):
        for spread in (0.1, 5.0, 800.0):
            v = rng.standard_normal(n) * spread
            if n > 2:
                v[n // 2] = v.max()                       # a repeated maximum: only the first one is left out of the sum
            norms.append({"v": hx(v, "<f8"), "out": hx([ref_log_exp_norm(v)], "<f8")})
    norms.append({"v": hx([-1.7976931348623157e+308, 0.0, -3.0], "<f8"),
                  "out": hx([ref_log_exp_norm([-1.7976931348623157e+308, 0.0, -3.0])], "<f8")})
    minus = []
    for dim in (1, 7, 40):
        x = np.abs(rng.standard_normal(dim)) * 1000 + 50
        y = np.abs(rng.standard_normal(dim)) * 40
        w = float(rng.uniform(1, 300))
        minus.append({"x": hx(x, "<f8"), "y": hx(y, "<f8"), "weight": hx([w], "<f8"), "out": hx(ref_normalized_minus(x, y, w), "<f4")})
    out = {"source": "oracle/_ref/libref.so: ref_accumulate_vector / ref_log_exp_norm / ref_normalized_minus (reference Mm/Utilities.hh "
                     "templates: unrolledTransform, plusWeighted, plusSquare, plusSquareWeighted, plusNormalizedSquare, normalizedMinus, logExpNorm)",
           "accumulate": acc, "log_exp_norm": norms, "normalized_minus": minus}
    with open(os.path.join(HERE, "ref_estimate.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote ref_estimate.json")


if __name__ == "__main__":
    main()
