"""tools/fuzz_backend.py [n_cases] [seed] -- random segmentations, dimensions, strides and window settings through the feature back-end
kernels (normalisation, regression, matrix multiplication) against oracle/orc_backend.c, bit for bit.  Every case draws one of the
reference's two arithmetics (amx_set_contract off | fma) and is held to the oracle library of that build."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle.binding import oracle_matrix_multiply, oracle_normalize, oracle_regression  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
ctx = rasr_amd.Context(0)
ctx.use_torch_stream()
fe = rasr_amd.MfccExtractor(ctx)
bad = 0
for case in range(n_cases):
    lens = [int(rng.choice([1, 2, 3, 7, 20, 63, 64, 65, 300, 1000])) for _ in range(int(rng.integers(1, 8)))]
    samples = [400 + 160 * (n - 1) for n in lens]                      # n frames each
    plan = fe.plan(np.concatenate([[0], np.cumsum(samples)]))
    off = np.concatenate([[0], np.cumsum(lens)])
    F = int(off[-1])
    assert plan.total_frames == F
    contract = ("off", "fma")[int(rng.integers(0, 2))]
    ctx.set_contract(contract)
    dim, pad = int(rng.integers(1, 70)), int(rng.integers(0, 5))
    ld = dim + pad
    x = (rng.standard_normal((F, ld)) * 3 + 1).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    segs = [x[off[i]:off[i + 1], :dim] for i in range(len(lens))]

    def check(what, got, fn):
        global bad
        for i, s in enumerate(segs):
            want = fn(s)
            if not np.array_equal(got[off[i]:off[i + 1]].view(np.uint32), want.view(np.uint32)):
                bad += 1
                print("MISMATCH", what, "contract", contract, "case", case, "lens", lens, "dim", dim, "segment", i)
                return

    kw = dict(variance=bool(rng.integers(0, 2)))
    if rng.integers(0, 2):
        L = int(rng.integers(1, 40))
        kw.update(length=L, right=int(rng.integers(0, L)))
    out = torch.zeros((F, dim), dtype=torch.float32, device="cuda")
    ctx.normalize(plan, xd, ld, dim, out, dim, **kw)
    torch.cuda.synchronize()
    check("normalize %s" % kw, out.cpu().numpy(), lambda s: oracle_normalize(s, **kw))
    order, right = int(rng.integers(1, 3)), int(rng.integers(1, 6))
    out = torch.zeros((F, dim), dtype=torch.float32, device="cuda")
    ctx.regression(plan, xd, ld, dim, out, dim, order=order, right=right)
    torch.cuda.synchronize()
    check("regression %d/%d" % (order, right), out.cpu().numpy(), lambda s: oracle_regression(s, order, right, contract=contract))
    rows = int(rng.integers(1, 60))
    M = rng.standard_normal((rows, dim)).astype(np.float32)
    out = torch.zeros((F, rows), dtype=torch.float32, device="cuda")
    ctx.matrix_multiply(torch.from_numpy(M).cuda(), rows, dim, xd, ld, F, out, rows)
    torch.cuda.synchronize()
    want = oracle_matrix_multiply(M, x[:, :dim], contract=contract)
    if not np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)):
        bad += 1
        print("MISMATCH matrix multiply case", case, rows, dim, F)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
