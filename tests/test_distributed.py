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


def _train_statistics(utts, model):
    """E-step of one rank: Baum-Welch statistics of its utterances (oracle as the scorer; frames aligned to their best state)"""
    from oracle import OracleGmm, OracleMfcc
    fe = OracleMfcc(n_ceps=12)
    gmm = OracleGmm(model)
    acc = np.zeros(gmm.accumulator_size())
    for u in utts:
        x = fe.run(synth.waveform(2000 + 37 * u, seed=500 + u))
        mix = gmm.score(x, want_best=False).argmin(axis=1).astype(np.uint32)
        gmm.accumulate_weighted(1, x, mix, None, None, acc)
    return acc


def _train_worker(rank, world, port, n_utt, q):
    import torch
    import torch.distributed as dist

    from rasr_amd.partition import select_partition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = synth.gmm_cart(8, 2, 4, 12, seed=3, pooled=False)
    acc = torch.from_numpy(_train_statistics(select_partition(n_utt, world, rank), model))
    dist.all_reduce(acc)                                   # the ONE exchange of an epoch: the flat f64 statistics
    if rank == 0:
        q.put(acc.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_epoch_equals_single_process():
    """accumulate on two ranks -> all-reduce -> M-step (amx_gmm_estimate, host code of the product) gives the model of the
    single-process epoch: statistics to 1e-12 (f64 summation order), estimated topology identical, parameters to f32 rounding"""
    import torch.multiprocessing as mp

    import rasr_amd
    n_utt = 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, n_utt, q)) for r in range(2)]
    for p in procs:
        p.start()
    combined = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = synth.gmm_cart(8, 2, 4, 12, seed=3, pooled=False)
    single = _train_statistics(range(n_utt), model)
    assert np.allclose(combined, single, rtol=1e-12, atol=1e-9)
    a = rasr_amd.gmm_estimate(model, combined, min_observation_weight=1.0, allow_zero_weights=1)
    b = rasr_amd.gmm_estimate(model, single, min_observation_weight=1.0, allow_zero_weights=1)
    for k in ("mix_offsets", "dens_index", "dens_mean", "dens_cov"):
        assert np.array_equal(a[k], b[k])
    assert np.allclose(a["means"], b["means"], rtol=1e-6, atol=1e-7) and np.allclose(a["variances"], b["variances"], rtol=1e-5)
    assert np.allclose(a["log_weight"], b["log_weight"], rtol=1e-10, atol=1e-12)
