"""GPU: the C++ host-side mirror (rasr_amd/host/*.hh) of the reference's buffered FeatureScorer protocol and of the
MFCC Flow node, compiled with g++ against librasr_amd.so, i.e. the same way a RASR adapter links it."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_protocol(tmp_path):
    exe = str(tmp_path / "host_protocol_test")
    lib = os.path.join(ROOT, "rasr_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host_protocol_test.cc"), "-o", exe,
                           "-L" + lib, "-lrasr_amd", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    dump = str(tmp_path / "ceps.bin")
    out = subprocess.run([exe, dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
    # the packets the node emitted equal the oracle chain on the same samples
    from oracle import OracleMfcc
    raw = np.fromfile(dump, dtype=np.float32)
    ceps, pcm = raw[:99 * 12].reshape(99, 12), raw[99 * 12:]
    want = OracleMfcc(n_ceps=12).run(pcm)
    assert np.all(np.abs(ceps - want) <= 1e-4 * np.abs(want) + 2e-3)


def test_host_epoch_reduce_client(tmp_path):
    """rasr_amd/host/EpochReduce.hh compiled with g++ against librasr_amd.so only (no torch, no Python in the process): communicator id
    through a file, ONE amx_comm_all_reduce_f64_dev over statistics + counters, results checked inside the C++ program.  One rank per
    GPU: as many ranks as the box has GPUs (1 on the test box)."""
    import torch
    exe = str(tmp_path / "host_epoch_reduce_test")
    lib = os.path.join(ROOT, "rasr_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host_epoch_reduce_test.cc"), "-o", exe,
                           "-L" + lib, "-lrasr_amd", "-lpthread", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    world = min(torch.cuda.device_count(), 8)
    idf = str(tmp_path / "comm.id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(r), str(world), idf], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(world)]
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and out.strip().endswith("OK"), "rank %d: %s" % (r, out)


def test_gather_refuses_pairs_outside_the_resident_block(ctx):
    """amx_gather_scores validates the decoder's (row, emission) pairs against the block's shape instead of reading wherever they point"""
    import torch

    import rasr_amd
    ctx.use_torch_stream()
    block = torch.arange(6 * 50, dtype=torch.float32, device="cuda").reshape(6, 50)
    got = ctx.gather_scores(block, 50, [0, 5, 3], [0, 49, 7])
    assert got.tolist() == [0.0, 5 * 50 + 49.0, 3 * 50 + 7.0]
    for rows, cols in (([6], [0]), ([0], [50]), ([0, 2 ** 31], [1, 1])):
        with pytest.raises(rasr_amd.AmxError) as e:
            ctx.gather_scores(block, 50, rows, cols)
        assert "outside the 6 x 50 block" in str(e.value)


def test_copy_to_device_from_pinned_memory_may_reuse_the_source_on_return(ctx):
    """amx_copy_to_device's contract ("the source may be reused on return") also for a pinned source, where hipMemcpyAsync alone
    would return before the DMA engine has read it"""
    import ctypes as C

    import torch

    from rasr_amd import _lib
    L = _lib.lib()
    n = 8 << 20
    src = torch.ones(n, dtype=torch.float32).pin_memory()
    dst = torch.zeros(n, dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    for rep in range(3):
        src.fill_(float(rep + 1))
        _lib.check(L.amx_copy_to_device(ctx.h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n * 4))
        src.fill_(-1.0)   # overwrite immediately: must not reach the device copy
        torch.cuda.synchronize()
        assert float(dst.min()) == float(dst.max()) == float(rep + 1)
