"""tools/mx_timeline.py [variant bits] -- where the time of a K-tile goes in gemm_mx_kernel (lab build of the library:
make -C rasr_amd/csrc OBJDIR=build_lab OUT=../../tools/build/librasr_amd_lab.so EXTRA=-DAMX_LAB).

Runs the config-5 output layer (2048 -> 10000, 32768 frames, f16mx) with DBG 2048 | variant bits: workgroup 0 stamps s_memtime in
every wave for its first 48 K-tiles -- 0 barrier passed, 1 refill issued, 2 fragments in registers (an extra lgkmcnt(0): the
instrumented kernel serialises reads and products of a wave), 3 products issued -- and prints per wave the mean cycles of the phases
and of the whole period (steady state: K-tiles 8..47).  a tick of s_memtime is one shader cycle (guide, constants table)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AMX_LIBRARY", os.path.join(ROOT, "tools", "build", "librasr_amd_lab.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rasr_amd  # noqa: E402
from rasr_amd import _lib  # noqa: E402
from tests import synth  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "small" else 0
small = len(sys.argv) > 1 and sys.argv[1] == "small"   # a 2048 x 2048 hidden layer at batch 1024: the 128 x 64 tiles, one per CU
ctx = rasr_amd.Context(0)
ctx.use_torch_stream()
Ws, bs, acts, logp = synth.ffnn([2048, 2048, 64] if small else [2048, 10000], seed=7)
T = 1024 if small else 32768
x = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).standard_normal((T, 2048)).astype(np.float32)).cuda()
nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx", tuning="graph=0,mx_dbg=%d" % (2048 | bits) + ("," + sys.argv[2] if len(sys.argv) > 2 else ""))
s = torch.empty((T, 64 if small else 10000), dtype=torch.float32, device="cuda")
for _ in range(3):
    nn.score_dev(x, 2048, T, s)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (8 * 48 * 4))()
L.amx_lab_mx_stamps.restype = C.c_int
assert L.amx_lab_mx_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8, 48, 4).astype(np.int64)
lo, hi = (4, 30) if small else (8, 47)
t0 = st[:, lo, 0][st[:, lo, 0] > 0].min()
print("variant bits %d%s; shader cycles (s_memtime); iterations %d..%d of workgroup 0" % (bits, " (small: 128 x 64 tiles, U K-tiles per iteration)" if small else "", lo, hi))
print("wave  period   barrier->refill  refill->frags  frags->products  products->next barrier   first barrier (rel)")
for w in range(8):
    if st[w, lo, 0] == 0:
        continue
    a = st[w, lo:hi]
    nxt = st[w, lo + 1:hi + 1, 0]
    per = np.diff(st[w, lo:hi + 1, 0]).mean()
    if (a[:, 2] == 0).all():   # a loop without the "fragments in registers" stamp (the read-ahead loop: reads ride between the products)
        print("%4d  %6.1f   %15.1f  %13s  %15.1f  %22.1f   %d   (refill -> products: reads and products interleaved)"
              % (w, per, (a[:, 1] - a[:, 0]).mean(), "-", (a[:, 3] - a[:, 1]).mean(), (nxt - a[:, 3]).mean(), st[w, lo, 0] - t0))
        continue
    print("%4d  %6.1f   %15.1f  %13.1f  %15.1f  %22.1f   %d" % (w, per, (a[:, 1] - a[:, 0]).mean(), (a[:, 2] - a[:, 1]).mean(),
                                                               (a[:, 3] - a[:, 2]).mean(), (nxt - a[:, 3]).mean(), st[w, lo, 0] - t0))

# tile level (workgroup 0, thread 0): tile start -> K-loop end -> tile end, for the workgroup's first four tiles; s_memrealtime
# (100 MHz) beside s_memtime gives the shader clock under this load
tb = (C.c_ulonglong * 24)()
L.amx_lab_mx_tile_stamps.restype = C.c_int
if L.amx_lab_mx_tile_stamps(tb) == 0:
    ts = np.frombuffer(tb, dtype=np.uint64).reshape(4, 3, 2).astype(np.int64)
    dc, dr = ts[-1, 2, 0] - ts[0, 0, 0], ts[-1, 2, 1] - ts[0, 0, 1]
    if ts[0, 0, 0] > 0 and dc > 0 and dr > 0:   # the small-batch tile carries no tile stamps (one tile per workgroup): nothing to print
        ghz = dc / (dr * 10.0)
        print("tiles of workgroup 0: %d s_memtime ticks in %d s_memrealtime ticks of 10 ns -> %.3f ticks per ns" % (dc, dr, ghz))
        print("tile   K-loop (prologue + %d K-tiles)   epilogue (+ the closing barrier)   gap to the next tile start      [us]" % (2048 // 32))
        for i in range(4):
            gap = (ts[i + 1, 0, 0] - ts[i, 2, 0]) / ghz / 1e3 if i < 3 else float("nan")
            print("%4d   %10.2f %30.2f %34.2f" % (i, (ts[i, 1, 0] - ts[i, 0, 0]) / ghz / 1e3, (ts[i, 2, 0] - ts[i, 1, 0]) / ghz / 1e3, gap))
