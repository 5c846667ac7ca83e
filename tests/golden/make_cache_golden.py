"""Regenerates tests/golden/ref_cache.json.  Run in the build container only (needs oracle/_ref/libref.so):

    python tests/golden/make_cache_golden.py

The expected bytes are OUTPUTS OF THE REFERENCE ITSELF: Flow::Vector<f32>::write and
Flow::Datatype::writeGatheredData over Core::BinaryOutputStream, and Core::XmlWriter, compiled unmodified
into libref.so (oracle/ref/ref_harness.cc: ref_cache_block_write, ref_attribs_xml).  Only inputs and the
expected output bytes are stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.binding import load_ref, ref_attribs_xml, ref_cache_block  # noqa: E402


def main():
    if load_ref() is None:
        raise SystemExit("oracle/_ref/libref.so not available (needs /root/reference)")
    rng = np.random.Generator(np.random.PCG64(77))
    cases = []
    for n, dim in ((0, 16), (1, 1), (5, 16), (3, 0), (12, 40)):
        x = (rng.standard_normal((n, dim)) * 7).astype(np.float32)
        if n:
            x.flat[:1] = np.float32(-0.0)
        start = np.arange(n) * 0.01 + 0.0125
        t = np.stack([start, start + 0.025], 1) if n else np.zeros((0, 2))
        cases.append({"n": n, "dim": dim, "feats_hex": x.tobytes().hex(), "times_hex": t.astype("<f8").tobytes().hex(),
                      "block_hex": ref_cache_block(x, t).hex()})
    attrs = [{"sample-rate": "100", "datatype": "vector-f32", "frame-shift": "0.01"},
             {"we<ird&\"q'": "a>b & \"c\" 'd'", "empty": ""}, {}]
    out = {"source": "oracle/_ref/libref.so: ref_cache_block_write / ref_attribs_xml (reference Flow::Vector<f32>, "
                     "Flow::Datatype, Core::BinaryOutputStream, Core::XmlWriter)",
           "blocks": cases, "attributes": [{"attrs": a, "xml": ref_attribs_xml(a)} for a in attrs]}
    with open(os.path.join(HERE, "ref_cache.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote ref_cache.json")


if __name__ == "__main__":
    main()
