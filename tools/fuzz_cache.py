"""tools/fuzz_cache.py [n_rounds] [seed] -- random sequences of write / overwrite / remove / reopen on SP_ARC1 archives through the C ABI
(host code, no GPU needed), checked against an in-memory model and against the independent parser in oracle/cache_format.py."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle import cache_format as cf  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
bad = 0
with tempfile.TemporaryDirectory() as d:
    for r in range(rounds):
        path = os.path.join(d, "a%d.cache" % r)
        model = {}
        a = rasr_amd.FileArchive(path, "w")
        for step in range(int(rng.integers(5, 60))):
            op = rng.integers(0, 10)
            name = "seg/%d" % rng.integers(0, 12)
            if op < 6:
                n = int(rng.choice([0, 1, 7, 100, 5000, 70000]))
                data = rng.integers(0, 256, n, dtype=np.uint8).tobytes() if rng.integers(0, 2) else bytes(n)
                a.write_file(name, data, compress=bool(rng.integers(0, 2)))
                model[name] = data
            elif op < 8 and model:
                victim = list(model)[int(rng.integers(0, len(model)))]
                a.remove_file(victim)
                del model[victim]
            else:
                a.close()
                a = rasr_amd.FileArchive(path, "w")
            if sorted(f[0] for f in a.files()) != sorted(model):
                bad += 1
                print("MISMATCH file list", r, step, sorted(f[0] for f in a.files()), sorted(model))
                break
        a.close()
        ro = rasr_amd.FileArchive(path, "r")
        for name, data in model.items():
            if ro.read_file(name) != data:
                bad += 1
                print("MISMATCH content", r, name, len(data))
        ro.close()
        files, _ = cf.parse_archive(open(path, "rb").read())
        if {k: v for k, v in files.items()} != model:
            bad += 1
            print("MISMATCH independent parser", r, sorted(files), sorted(model))
print("%d archives, %d mismatches" % (rounds, bad))
sys.exit(1 if bad else 0)
