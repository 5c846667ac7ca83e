"""CPU, world_size 2 over gloo: utterance sharding follows the reference's partition rule and the epoch
all-reduce of the accumulators equals the single-process result."""
import os
import socket

import numpy as np
import pytest

from tests import synth


def test_partition_rule():
    from rasr_amd.partition import select_partition
    assert list(select_partition(7, 0, 0)) == list(range(7))
    assert list(select_partition(7, 2, 0)) == [0, 2, 4, 6]
    assert list(select_partition(7, 2, 1)) == [1, 3, 5]
    assert list(select_partition(7, 2, 2)) == [0, 2, 4, 6]      # select == partition means partition 0
    with pytest.raises(ValueError):
        select_partition(7, 2, 3)
    parts = [select_partition(1000, 8, k) for k in range(8)]
    assert sorted(np.concatenate(parts)) == list(range(1000))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _accumulate(utts, model_seed, n_states):
    """what one rank does per utterance: features -> scores -> best state -> accumulators (oracle as the scorer)"""
    import torch

    from oracle import OracleGmm, OracleMfcc
    from rasr_amd.partition import EpochAccumulators
    acc = EpochAccumulators(n_states)
    fe = OracleMfcc(n_ceps=12)
    gmm = OracleGmm(synth.gmm_cart(n_states, 1, 3, 12, seed=model_seed))
    for u in utts:
        x = fe.run(synth.waveform(2000 + 37 * u, seed=500 + u))
        sc = gmm.score(x, want_best=False)
        best = sc.argmin(axis=1)
        acc.counts += torch.from_numpy(np.bincount(best, minlength=n_states))
        acc.score_sum += float(sc[np.arange(len(best)), best].astype(np.float64).sum())
        acc.n_frames += len(best)
    return acc


def _worker(rank, world, port, n_utt, n_states, q):
    import torch.distributed as dist

    from rasr_amd.partition import select_partition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = _accumulate(select_partition(n_utt, world, rank), 3, n_states)
    acc.all_reduce()
    if rank == 0:
        q.put((acc.counts.numpy().copy(), float(acc.score_sum[0]), int(acc.n_frames[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_epoch_reduce_equals_single_process():
    import torch.multiprocessing as mp
    n_utt, n_states = 9, 20
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, n_states, q)) for r in range(2)]
    for p in procs:
        p.start()
    counts, ssum, nfr = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _accumulate(range(n_utt), 3, n_states)
    assert np.array_equal(counts, single.counts.numpy())
    assert nfr == int(single.n_frames[0]) == int(counts.sum())
    assert abs(ssum - float(single.score_sum[0])) <= 1e-12 * abs(ssum)   # f64 sums, order differs
