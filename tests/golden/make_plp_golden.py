"""Regenerates tests/golden/ref_levinson.json.  Run in the build container only (needs oracle/_ref/libref.so):

    python tests/golden/make_plp_golden.py

Expected values are OUTPUTS OF THE REFERENCE ITSELF: Math::LevinsonLeastSquares (src/Math/LevinsonLse.cc, compiled unmodified into
libref.so) driven like Signal::AutocorrelationToAutoregressionNode::work (oracle/ref/ref_harness.cc: ref_levinson).  Only inputs
(autocorrelation sequences, f32) and expected outputs (gain and a1..aN as f32, or the failure flag) are stored, as hex.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.binding import load_ref, ref_levinson  # noqa: E402


def main():
    if load_ref() is None:
        raise SystemExit("oracle/_ref/libref.so not available (needs /root/reference)")
    rng = np.random.Generator(np.random.PCG64(17))
    cases = []
    for trial in range(60):
        n = int(rng.integers(2, 33))
        x = rng.standard_normal(512) * (10.0 ** rng.uniform(-4, 4))
        R = np.array([np.dot(x[:512 - k], x[k:]) for k in range(n)], np.float32)
        if trial % 9 == 0:
            R[1:] = 0                                    # white: all reflection coefficients zero
        if trial == 5:
            R[:] = 0                                     # digital silence: the recursion fails
        if trial == 6:
            R = np.full(n, R[0], np.float32)             # perfectly predictable: prediction error collapses to zero
        out = ref_levinson(R)
        cases.append({"R": R.astype("<f4").tobytes().hex(),
                      "ok": out is not None,
                      "gain": "" if out is None else np.float32(out[0]).tobytes().hex(),
                      "a": "" if out is None else out[1].astype("<f4").tobytes().hex()})
    with open(os.path.join(HERE, "ref_levinson.json"), "w") as f:
        json.dump({"source": "oracle/_ref/libref.so: ref_levinson (reference Math/LevinsonLse.cc, unmodified)", "cases": cases}, f, indent=1)
    print("wrote ref_levinson.json")


if __name__ == "__main__":
    main()
