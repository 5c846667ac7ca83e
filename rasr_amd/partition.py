"""Corpus partitioning and the per-epoch accumulator reduce (host logic, no device code).

RASR has no communication backend: data-parallel jobs are independent processes that each take
`*.corpus.partition = N`, `select-partition = k` (segment i belongs to partition i % N,
Bliss/CorpusDescription.cc:242-248,488-498; select-partition == N is accepted as 0) and write their own
accumulator files, which `combine-mixture-set-estimators` sums offline
(Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325).  Here the ranks of one torch.distributed job
take the same partitions and the sum is ONE all-reduce: amx_comm_all_reduce_f64_dev (RCCL over xGMI, include/amx.h) on GPUs,
torch.distributed's gloo backend in the CPU tests.
"""
import numpy as np


def select_partition(n_segments, partition, select):
    """indices of the segments rank `select` of `partition` processes visits (reference rule)."""
    if partition <= 0:
        return np.arange(n_segments)
    if select == partition:
        select = 0  # "This convention is useful for SGE array jobs"
    elif select > partition or select < 0:
        raise ValueError("Invalid partition %d (should be 0 - %d)." % (select, partition))
    return np.arange(select, n_segments, partition)


class CorpusWalker:
    """The order in which one rank of a data-parallel job visits a corpus: the segments of its partition (`select_partition`,
    i.e. the reference's `partition` / `select-partition` rule, Bliss/CorpusDescription.cc:174-190), `batch` at a time, in corpus
    order (the reference's CorpusVisitor walks recordings / segments in document order, Speech/CorpusProcessor.cc:49-58).
    A partition that is not a multiple of `batch` ends with a short batch; `next_batch` starts the next epoch after it.

    Host logic only: the streamed ingest of bench.py asks it which utterances to copy next, and the gloo test checks that two
    ranks walk disjoint lists that cover the corpus."""

    def __init__(self, n_segments, partition, select, batch):
        if batch <= 0:
            raise ValueError("batch must be positive")
        self.segments = select_partition(n_segments, partition, select)
        if len(self.segments) == 0:
            raise ValueError("partition %d of %d holds no segment of a %d-segment corpus" % (select, partition, n_segments))
        self.batch = int(batch)
        self.pos = 0
        self.epoch = 0

    def batches_per_epoch(self):
        return (len(self.segments) + self.batch - 1) // self.batch

    def next_batch(self):
        """corpus indices of the next `batch` segments of this rank (fewer at the end of its partition)"""
        if self.pos >= len(self.segments):
            self.pos = 0
            self.epoch += 1
        out = self.segments[self.pos:self.pos + self.batch]
        self.pos += len(out)
        return out


class EpochReduceBuffer:
    """Everything a rank contributes to the per-epoch sum, in ONE flat f64 buffer, so that the exchange is literally one
    all-reduce (one RCCL ring over xGMI per epoch; replaces `combine-mixture-set-estimators`,
    Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325).

    fields: list of (name, n, kind); kind "f64" = a statistics block the kernels accumulate into IN PLACE (view(name) is a
    slice of the flat buffer: the 53.8 MB of GMM statistics are never copied), kind "count" = an int64 / u64 counter tensor
    the kernels update with integer atomics; counters travel as f64 in the tail of the buffer (exact below 2^53 -- a 100 h
    corpus has 3.6e7 frames) and are written back as integers after the reduce.
    """

    def __init__(self, fields, device="cpu"):
        import torch
        self.fields = list(fields)
        self.offsets, off = {}, 0
        for name, n, kind in self.fields:
            if kind not in ("f64", "count"):
                raise ValueError("unknown field kind %r" % kind)
            self.offsets[name] = (off, n, kind)
            off += n
        self.flat = torch.zeros(off, dtype=torch.float64, device=device)
        self.counters = {name: torch.zeros(n, dtype=torch.int64, device=device) for name, n, kind in self.fields if kind == "count"}

    def view(self, name):
        """the tensor the kernels write: a slice of the flat buffer (f64 fields) or the integer counter tensor"""
        off, n, kind = self.offsets[name]
        return self.flat[off:off + n] if kind == "f64" else self.counters[name]

    def nbytes(self):
        return int(self.flat.numel() * 8)

    def zero(self):
        """start of an epoch: statistics and counters back to zero (in place: the kernels keep their views)"""
        self.flat.zero_()
        for c in self.counters.values():
            c.zero_()
        return self

    def all_reduce(self, group=None, comm=None):
        """sum over all ranks with ONE collective.

        comm (rasr_amd.Comm): the product path -- amx_comm_all_reduce_f64_dev on the flat device buffer (RCCL over xGMI), counters
        converted on the device by amx_counts_to_f64_dev / amx_f64_to_counts_dev.  Without one: torch.distributed (the gloo tests
        on CPU), or a no-op when no process group is initialised."""
        import torch
        if comm is not None:
            for name, c in self.counters.items():
                off, n, _ = self.offsets[name]
                comm.counts_to_f64(c, self.flat[off:off + n])
            comm.all_reduce_f64(self.flat)
            for name, c in self.counters.items():
                off, n, _ = self.offsets[name]
                comm.f64_to_counts(self.flat[off:off + n], c)
            return self
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return self
        for name, c in self.counters.items():
            off, n, _ = self.offsets[name]
            self.flat[off:off + n] = c.to(torch.float64)
        dist.all_reduce(self.flat, group=group)
        for name, c in self.counters.items():
            off, n, _ = self.offsets[name]
            c.copy_(self.flat[off:off + n].round().to(torch.int64))
        return self


class EpochAccumulators(EpochReduceBuffer):
    """state_counts u64[M] (kept as int64 tensors), score_sum f64[1], n_frames i64[1] on `device`."""

    def __init__(self, n_states, device="cpu"):
        super().__init__([("score_sum", 1, "f64"), ("counts", n_states, "count"), ("n_frames", 1, "count")], device)
        self.counts = self.view("counts")
        self.score_sum = self.view("score_sum")
        self.n_frames = self.view("n_frames")
