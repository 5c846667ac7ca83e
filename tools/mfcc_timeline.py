"""tools/mfcc_timeline.py -- where a tile's time goes in mfcc_kernel (lab build: make -C rasr_amd/csrc OBJDIR=build_lab
OUT=../../tools/build/librasr_amd_lab.so EXTRA=-DAMX_LAB CHECK=-).  Workgroup 0 of the config-2 run (1000 utterances, MFCC-40) stamps s_memtime per
wave and tile: tile start, phase B (4 frames per wave: samples -> FFT -> amplitudes) done, barrier, phase C (mel filter bank + log) done,
barrier, phase D (DCT on the matrix cores + stores) done, barrier.  Prints per wave the mean share of each interval over tiles 4..30."""
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
fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0, tuning=sys.argv[1] if len(sys.argv) > 1 else None)
lens = synth.utterance_lengths(1000, seed=3)
base = synth.waveform(int(lens.max()) + 1000, seed=4)
pcm = np.concatenate([base[u:u + int(n)] for u, n in enumerate(lens)])
off = np.concatenate([[0], np.cumsum(lens)])
plan = fe.plan(off)
x = torch.from_numpy(pcm).cuda()
out = torch.empty((plan.total_frames, 40), dtype=torch.float32, device="cuda")
for _ in range(3):
    fe.run_plan(plan, x, out)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (4 * 32 * 8))()
L.amx_lab_mfcc_stamps.restype = C.c_int
assert L.amx_lab_mfcc_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(4, 32, 8).astype(np.int64)
names = ["phase B (4 frames)", "barrier", "phase C (mel)", "barrier", "phase D (DCT, stores)", "barrier"]
print("s_memtime ticks, workgroup 0, tiles 4..30 (a tile = 16 frames)")
print("wave  tile period  " + "  ".join("%-22s" % n for n in names))
for w in range(4):
    a = st[w, 4:30]
    per = np.diff(st[w, 4:31, 0]).mean()
    ph = [(a[:, k + 1] - a[:, k]).mean() for k in range(6)]
    print("%4d  %11.0f  " % (w, per) + "  ".join("%8.0f (%4.1f %%)       " % (v, 100 * v / per) for v in ph))

if len(sys.argv) > 1 and "r16" in sys.argv[1]:  # inside the radix-16 phase B (one batch of four frames per wave)
    b2 = (C.c_ulonglong * (4 * 32 * 8))()
    L.amx_lab_mfcc_r16_stamps.restype = C.c_int
    assert L.amx_lab_mfcc_r16_stamps(b2) == 0
    r = np.frombuffer(b2, dtype=np.uint64).reshape(4, 32, 8).astype(np.int64)
    print("radix-16 phase B: tile start -> samples, pre-emphasis, window | DFT-16 + twiddle | transposition | DFT-16 + natural-order write | split, amplitudes")
    for w in range(4):
        a, t0, t1 = r[w, 4:30], st[w, 4:30, 0], st[w, 4:30, 1]
        ok = (a[:, :4] > 0).all(axis=1) & (a[:, 0] >= t0) & (t1 >= a[:, 3])  # tiles in which this wave had a batch
        a, t0, t1 = a[ok], t0[ok], t1[ok]
        ph = [(a[:, 0] - t0).mean(), (a[:, 1] - a[:, 0]).mean(), (a[:, 2] - a[:, 1]).mean(), (a[:, 3] - a[:, 2]).mean(), (t1 - a[:, 3]).mean()]
        print("%4d  " % w + "  ".join("%8.0f" % v for v in ph))
