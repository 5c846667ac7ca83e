"""tools/fused_timeline.py -- where a tile's time goes in gmm_fused_kernel (lab build: make -C rasr_amd/csrc OBJDIR=build_lab
OUT=../../tools/build/librasr_amd_lab.so EXTRA=-DAMX_LAB).  Workgroup 0 of a 63 936-frame pass over the 10 000 x 16 model stamps s_memtime per wave
and tile: barrier passed, screen done, first survivors done, further survivors done, results issued.  Prints per wave the mean
cycles of the phases over tiles 8..55 and the wait at the next barrier (= period - busy)."""
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

ctx = rasr_amd.Context(0)
ctx.use_torch_stream()
model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=sys.argv[1] if len(sys.argv) > 1 else None)
T = 63936
x = torch.from_numpy(np.random.Generator(np.random.PCG64(4)).standard_normal((T, 40)).astype(np.float32)).cuda()
s = torch.empty((T, 10000), dtype=torch.float32, device="cuda")
b = torch.empty((T, 10000), dtype=torch.int32, device="cuda")
for _ in range(2):
    sc.score_dev(x, T, s, b)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (16 * 64 * 6))()
L.amx_lab_fused_stamps.restype = C.c_int
assert L.amx_lab_fused_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(16, 64, 6).astype(np.int64)
print("shader cycles (s_memtime), workgroup 0, tiles 8..55")
print("wave  period   screen   first survivors   further survivors   results   wait at next barrier")
for w in range(16):
    a = st[w, 8:56]
    if a[0, 0] == 0:
        continue
    per = np.diff(st[w, 8:57, 0]).mean()
    ph = [(a[:, k + 1] - a[:, k]).mean() for k in range(4)]
    print("%4d  %6.1f  %7.1f  %16.1f  %18.1f  %8.1f  %21.1f" % (w, per, ph[0], ph[1], ph[2], ph[3], per - sum(ph)))
fs = (st[:12, 8:56, 3] - st[:12, 8:56, 2])
print("further survivors: mean %.0f, per-tile max over waves %.0f (the barrier waits for the slowest wave)" % (fs.mean(), fs.max(axis=0).mean()))
tot = (st[:12, 8:56, 4] - st[:12, 8:56, 0])
print("busy per tile: mean over waves %.0f, max over waves %.0f" % (tot.mean(), tot.max(axis=0).mean()))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    sc.score_dev(x, T, s, b)
ev1.record()
torch.cuda.synchronize()
print("kernel + pack: %.3f ms per pass" % (ev0.elapsed_time(ev1) / 5))
