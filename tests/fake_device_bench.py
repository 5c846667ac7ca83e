"""tests/fake_device_bench.py -- TEST INFRASTRUCTURE: runs bench.py's main() -- the GPU branch, `--backend nccl`, the `pipeline`
workload -- in a process that has no GPU, with every DEVICE call replaced HERE by a host stand-in, so that the control flow of an
N-rank run (launcher -> rendezvous on 127.0.0.1 -> reference partition rule -> streamed corpus walk -> EpochReduceBuffer -> ONE
all-reduce -> rank 0's JSON line) is executed end to end by tests/test_distributed.py before the first real 8-GPU run.

Nothing in bench.py or rasr_amd/ knows about this file: the product keeps raising without a GPU (`amx_init: no HIP device visible`).
The stand-ins do no arithmetic of the product; they count frames so that the reduce can be checked:
  FakeContext / FakeMfcc / FakeGmm / FakeNn   the methods bench.py calls on rasr_amd.Context, MfccExtractor, GmmFeatureScorer,
                                               NnBatchFeatureScorer; an uncovered call raises AttributeError naming it
  FakeComm                                    amx_comm_*: the all-reduce goes through torch.distributed (gloo) and is COUNTED
  torch.cuda.* / device="cuda"                streams and events are no-ops, "cuda" tensors live on the CPU, "nccl" becomes "gloo"
The launcher is bench.py's own (`launch_ranks`): bench.__file__ is pointed at this file, so the ranks it starts come up with the same
stand-ins.  usage: python tests/fake_device_bench.py <bench.py arguments>"""
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

COLLECTIVES = {"all_reduce_f64": 0}


class FakeStream:
    def __init__(self, *a, **k):
        pass

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def synchronize(self):
        pass


class FakeEvent:
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 1.0


class Strict:
    """an uncovered device call fails the test with its name (the stand-ins must follow bench.py, not hide a new call)"""

    def __getattr__(self, name):
        raise AttributeError("tests/fake_device_bench.py: bench.py called %s.%s, which has no stand-in here" % (type(self).__name__, name))


class FakeContext(Strict):
    def __init__(self, device=0):
        self.device = device

    def use_torch_stream(self):
        pass

    def set_contract(self, contract):
        assert contract in ("off", "fma")
        return self

    def profile(self, on):
        pass

    def profile_reset(self):
        pass

    def profile_get(self, key):
        return 0.0, 0

    def context_window(self, plan, ceps, dim, left, right, out, ld):
        pass

    def device_clocks(self, t):
        pass

    def device_clocks_xcd(self, t):
        pass

    def close(self):
        pass


class FakePlan:
    def __init__(self, offsets):
        n = np.diff(np.asarray(offsets, dtype=np.int64))
        # Signal/WindowBuffer.cc: ceil((N - 400) / 160) + 1 frames for N > 400
        self.frames = np.where(n > 400, -(-(n - 400) // 160) + 1, (n > 0).astype(np.int64))
        self.total_frames = int(self.frames.sum())


class FakeMfcc(Strict):
    def __init__(self, ctx, **cfg):
        pass

    def plan(self, offsets):
        return FakePlan(offsets)

    def run_plan(self, plan, pcm, ceps):
        assert pcm.dtype in (torch.int16, torch.float32)


class FakeNn(Strict):
    def __init__(self, ctx, Ws, bs, acts, **k):
        self.M = int(Ws[-1].shape[0])

    def effective_precision(self):
        return "f16mx", 2.4

    def score_stats_dev(self, x, ld, T, scores, best, counts, score_sum):
        counts[torch.arange(T) % self.M] += 1   # one frame per call and row: the reduce must find every frame of every rank
        score_sum += 0.5 * T


class FakeGmm(Strict):
    def __init__(self, ctx, model, tuning=None):
        self.M = len(model["mix_offsets"]) - 1

    def accumulator_size(self):
        return 64

    def score_stats_dev(self, x, T, scores, best_density, state, counts, score_sum):
        counts[torch.arange(T) % self.M] += 1
        score_sum += 0.25 * T

    def best_density_dev(self, x, T, state, out):
        pass

    def accumulate_dev(self, x, T, state, best, ld, acc):
        acc[0] += T

    def screen_counts(self, reset):
        return 0, 0


class FakeComm(Strict):
    def __init__(self, ctx, rank, world, uid):
        assert uid == b"fake-unique-id"
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        return b"fake-unique-id"

    def counts_to_f64(self, c, f):
        f.copy_(c.to(torch.float64))

    def f64_to_counts(self, f, c):
        c.copy_(f.round().to(torch.int64))

    def all_reduce_f64(self, flat):
        COLLECTIVES["all_reduce_f64"] += 1
        dist.all_reduce(flat)

    def close(self):
        pass


def _cpu_device(kw):
    d = kw.get("device")
    if d is not None and "cuda" in str(d):
        kw["device"] = "cpu"
    return kw


def install():
    import rasr_amd
    rasr_amd.Context = FakeContext
    rasr_amd.MfccExtractor = FakeMfcc
    rasr_amd.NnBatchFeatureScorer = FakeNn
    rasr_amd.GmmFeatureScorer = FakeGmm
    rasr_amd.Comm = FakeComm
    rasr_amd.version = lambda: "fake device (tests/fake_device_bench.py)"
    for name in ("empty", "zeros", "ones", "tensor", "full", "empty_like", "zeros_like"):
        orig = getattr(torch, name)
        setattr(torch, name, (lambda o: lambda *a, **k: o(*a, **_cpu_device(k)))(orig))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    cu = torch.cuda
    cu.device_count = lambda: 8
    cu.set_device = lambda d: None
    cu.synchronize = lambda *a, **k: None
    cu.empty_cache = lambda: None
    cu.Stream = FakeStream
    cu.Event = FakeEvent
    cu.stream = lambda s: contextlib.nullcontext()
    cu.current_stream = lambda *a, **k: FakeStream()
    init = dist.init_process_group

    def init_gloo(backend=None, **k):
        assert backend == "nccl", backend      # bench.py asks for RCCL; the stand-in process group is gloo
        k.pop("device_id", None)
        return init("gloo", **k)
    dist.init_process_group = init_gloo


def main():
    install()
    import bench
    bench.__file__ = os.path.abspath(__file__)   # bench.launch_ranks() starts the ranks with this file
    bench.main()
    if int(os.environ.get("RANK", "0")) == 0 and "WORLD_SIZE" in os.environ:
        print(json.dumps({"fake_device": True, "all_reduce_f64_calls_on_rank0": COLLECTIVES["all_reduce_f64"]}))


if __name__ == "__main__":
    main()
