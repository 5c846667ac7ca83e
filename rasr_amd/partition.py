"""Corpus partitioning and the per-epoch accumulator reduce (host logic, no device code).

RASR has no communication backend: data-parallel jobs are independent processes that each take
`*.corpus.partition = N`, `select-partition = k` (segment i belongs to partition i % N,
Bliss/CorpusDescription.cc:242-248,488-498; select-partition == N is accepted as 0) and write their own
accumulator files, which `combine-mixture-set-estimators` sums offline
(Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325).  Here the ranks of one torch.distributed job
take the same partitions and the sum is ONE all-reduce (RCCL over xGMI on GPUs, gloo in CPU tests).
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


class EpochAccumulators:
    """state_counts u64[M] (kept as int64 tensors), score_sum f64[1], n_frames i64[1] on `device`."""

    def __init__(self, n_states, device="cpu"):
        import torch
        self.counts = torch.zeros(n_states, dtype=torch.int64, device=device)
        self.score_sum = torch.zeros(1, dtype=torch.float64, device=device)
        self.n_frames = torch.zeros(1, dtype=torch.int64, device=device)

    def all_reduce(self, group=None):
        """sum over all ranks; no-op without an initialised process group"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.counts, group=group)
            dist.all_reduce(self.score_sum, group=group)
            dist.all_reduce(self.n_frames, group=group)
        return self
