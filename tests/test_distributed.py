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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_epoch_reduce_is_one_collective(monkeypatch):
    """EpochReduceBuffer: statistics blocks are views of the flat buffer (no copy), counters travel in its tail, and the
    exchange is exactly ONE all_reduce call (single-rank gloo group; the call count is what DESIGN.md section 8 promises)"""
    import torch
    import torch.distributed as dist

    from rasr_amd.partition import EpochReduceBuffer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        red = EpochReduceBuffer([("acc", 1000, "f64"), ("score_sum", 1, "f64"), ("counts", 50, "count"), ("gcounts", 50, "count")])
        acc = red.view("acc")
        assert acc.data_ptr() == red.flat.data_ptr() and red.nbytes() == 8 * (1000 + 1 + 50 + 50)
        acc += torch.arange(1000, dtype=torch.float64)
        red.view("score_sum").fill_(-3.5)
        red.view("counts")[7] = 2 ** 40 + 3
        red.view("gcounts")[49] = 11
        calls = []
        real = dist.all_reduce
        monkeypatch.setattr(dist, "all_reduce", lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1])
        red.all_reduce()
        assert calls == [1101]
        assert int(red.view("counts")[7]) == 2 ** 40 + 3 and int(red.view("gcounts")[49]) == 11 and int(red.view("counts").sum()) == 2 ** 40 + 3
        assert float(red.view("score_sum")[0]) == -3.5 and float(acc[999]) == 999.0
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_all_reduce_of_device_accumulators(ctx):
    """the RCCL path itself (backend "nccl", world size 1 on the one-GPU box): real device accumulators -- GMM Viterbi
    statistics, per-state counts and score sums produced by the HIP kernels -- go through ONE all-reduce and come back unchanged"""
    import torch
    import torch.distributed as dist

    import rasr_amd
    from rasr_amd.partition import EpochReduceBuffer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model = synth.gmm_cart(200, 1, 16, 40, seed=61, pooled=True)
        sc = rasr_amd.GmmFeatureScorer(ctx, model)
        T, M = 3000, 200
        x = np.random.Generator(np.random.PCG64(62)).standard_normal((T, 40)).astype(np.float32)
        red = EpochReduceBuffer([("acc", sc.accumulator_size(), "f64"), ("score_sum", 1, "f64"), ("counts", M, "count")], device="cuda")
        xd = torch.from_numpy(x).cuda()
        scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
        bestd = torch.empty((T, M), dtype=torch.int32, device="cuda")
        state = torch.empty((T,), dtype=torch.int32, device="cuda")
        ctx.use_torch_stream()
        sc.score_stats_dev(xd, T, scores, bestd, state, red.view("counts"), red.view("score_sum"))
        sc.accumulate_dev(xd, T, state, bestd, M, red.view("acc"))
        torch.cuda.synchronize()
        before = (red.view("acc").cpu().numpy().copy(), red.view("counts").cpu().numpy().copy(), float(red.view("score_sum").item()))
        assert before[1].sum() == T and before[0][:sc.accumulator_size()].sum() != 0
        calls = []
        real = dist.all_reduce

        def counted(t, *a, **k):
            calls.append((t.numel(), t.is_cuda))
            return real(t, *a, **k)
        dist.all_reduce = counted
        try:
            red.all_reduce()
        finally:
            dist.all_reduce = real
        torch.cuda.synchronize()
        assert calls == [(sc.accumulator_size() + 1 + M, True)]
        assert np.array_equal(red.view("acc").cpu().numpy(), before[0]) and np.array_equal(red.view("counts").cpu().numpy(), before[1])
        assert float(red.view("score_sum").item()) == before[2]
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- N ranks, started by bench.py itself

def _run_bench(argv, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE in the environment (the shape of the driver's 1-GPU command with N
    changed): bench.py starts the ranks, they meet on 127.0.0.1, take the reference's partitions, reduce once, and rank 0 reports
    n_gpus = 2.  Host-only stand-in for the rank's work (gloo): no GPU here."""
    rc, line, err = _run_bench(["--gpus", "2", "--backend", "gloo", "--workload", "null", "--steps", "3", "--warmup", "1", "--utterances", "9"])
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["stages"]["reduce_ok"] and line["stages"]["reduced_frames"] == line["stages"]["expected_frames"] > 0
    assert line["epoch_reduce"]["collectives"] == 1


def test_corpus_walker_batches_follow_the_partition_rule():
    """the streamed ingest's walk: batches of the rank's partition in corpus order, a short batch at its end, then the next epoch"""
    from rasr_amd.partition import CorpusWalker, select_partition
    w = CorpusWalker(23, 3, 1, 4)
    assert w.batches_per_epoch() == 2
    seen = [w.next_batch().tolist() for _ in range(3)]
    assert seen[0] == [1, 4, 7, 10] and seen[1] == [13, 16, 19, 22] and seen[2] == seen[0] and w.epoch == 1
    w = CorpusWalker(10, 4, 3, 2)
    assert [w.next_batch().tolist() for _ in range(3)] == [[3, 7], [3, 7], [3, 7]]
    w = CorpusWalker(10, 4, 1, 2)
    assert [w.next_batch().tolist() for _ in range(3)] == [[1, 5], [9], [1, 5]]
    parts = [CorpusWalker(1000, 8, k, 64) for k in range(8)]
    got = np.concatenate([np.concatenate([p.next_batch() for _ in range(p.batches_per_epoch())]) for p in parts])
    assert sorted(got.tolist()) == list(range(1000))
    assert all(np.array_equal(p.segments, select_partition(1000, 8, k)) for k, p in enumerate(parts))
    with pytest.raises(ValueError):
        CorpusWalker(3, 8, 5, 2)   # this partition holds no segment


def test_two_ranks_stream_disjoint_utterance_lists():
    """`bench.py --gpus 2` defaults to --ingest streamed: every step takes the NEXT utterances of the rank's partition of the 100 h
    corpus.  The host-only stand-in runs the same walk (CorpusWalker) on two gloo ranks; the lists are gathered over the control
    plane: disjoint, every index in its rank's partition, steps x batch x ranks utterances in total."""
    rc, line, err = _run_bench(["--gpus", "2", "--backend", "gloo", "--workload", "null", "--steps", "4", "--warmup", "1", "--utterances", "16"])
    assert rc == 0, err[-2000:]
    assert line["config"]["ingest"] == "streamed"
    walk = line["stages"]["corpus_walk"]
    assert walk["corpus_utterances"] == 36000 and walk["visited"] == walk["distinct"] == (4 + 1) * 16 * 2
    assert walk["ranks_disjoint"] and walk["every_rank_in_its_partition"]
    assert walk["first_of_each_rank"] == [[0, 2, 4], [1, 3, 5]]
    rc, line, err = _run_bench(["--backend", "gloo", "--workload", "null", "--steps", "2", "--warmup", "0"])
    assert rc == 0 and line["config"]["ingest"] == "resident" and "corpus_walk" not in line["stages"]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """ranks started by someone else with another world size than --gpus: fail loudly instead of printing a wrong n_gpus"""
    rc, line, err = _run_bench(["--gpus", "2", "--backend", "gloo", "--workload", "null"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and line is None and "WORLD_SIZE=1 but --gpus 2" in err


def test_bench_product_workloads_refuse_the_cpu_backend():
    rc, line, err = _run_bench(["--backend", "gloo", "--workload", "nn"])
    assert rc != 0 and line is None and "no CPU path" in err


# ---------------------------------------------------------------------------------------------- the product's host path on two ranks

def _cache_worker(rank, world, port, path, names, q):
    """one rank of a feature-cache pass: its partition of the segments, read through the product's archive reader (amx_feature_cache_read,
    host code of librasr_amd.so), per-dimension sums and frame counts into the flat reduce buffer, ONE all-reduce"""
    import torch
    import torch.distributed as dist

    import rasr_amd
    from rasr_amd.partition import EpochReduceBuffer, select_partition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    red = EpochReduceBuffer([("sum_x", 12, "f64"), ("sum_xx", 12, "f64"), ("frames", 1, "count"), ("segments", 1, "count")])
    ar = rasr_amd.FileArchive(path, "r")
    for i in select_partition(len(names), world, rank):
        x, _ = ar.read_features(names[i])
        x = x.astype(np.float64)
        red.view("sum_x").add_(torch.from_numpy(x.sum(axis=0)))
        red.view("sum_xx").add_(torch.from_numpy((x * x).sum(axis=0)))
        red.view("frames").add_(x.shape[0])
        red.view("segments").add_(1)
    ar.close()
    red.all_reduce()
    if rank == 0:
        q.put((red.view("sum_x").numpy().copy(), red.view("sum_xx").numpy().copy(), int(red.view("frames")[0]), int(red.view("segments")[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_read_their_partitions_of_a_feature_cache(tmp_path):
    """world size 2 over gloo with the PRODUCT's host code on both ranks: a SP_ARC1 feature cache written by amx_feature_cache_write,
    every rank reads the segments of its partition (i % 2) through amx_feature_cache_read and the reduced statistics equal the
    single-process pass over all segments"""
    import torch.multiprocessing as mp

    import rasr_amd
    rng = np.random.Generator(np.random.PCG64(77))
    path = str(tmp_path / "features.cache")
    names, feats = [], []
    ar = rasr_amd.FileArchive(path, "w")
    for u in range(7):
        n = 20 + 13 * u
        x = rng.standard_normal((n, 12)).astype(np.float32)
        t = np.stack([np.arange(n) * 0.01, np.arange(n) * 0.01 + 0.025], axis=1)
        name = "corpus/rec%d/seg%d" % (u // 3, u)
        ar.write_features(name, x, t, compress=bool(u & 1))
        names.append(name)
        feats.append(x)
    ar.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cache_worker, args=(r, 2, port, path, names, q)) for r in range(2)]
    for p in procs:
        p.start()
    sx, sxx, nfr, nseg = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    allx = np.concatenate(feats).astype(np.float64)
    assert nseg == 7 and nfr == allx.shape[0]
    assert np.allclose(sx, allx.sum(axis=0), rtol=1e-12, atol=1e-9) and np.allclose(sxx, (allx * allx).sum(axis=0), rtol=1e-12)


@pytest.mark.gpu
def test_amx_comm_all_reduce_through_the_c_abi(ctx):
    """amx_comm_* (RCCL bound by the library itself, no torch.distributed anywhere): communicator of one rank on the one-GPU box,
    real device accumulators of the HIP kernels through EpochReduceBuffer.all_reduce(comm=...) -- ONE amx_comm_all_reduce_f64_dev --
    and back unchanged, counters exact through their f64 slots"""
    import torch

    import rasr_amd
    from rasr_amd.partition import EpochReduceBuffer
    assert rasr_amd.Comm.available()
    uid = rasr_amd.Comm.unique_id()
    assert len(uid) == 128 and uid != bytes(128)
    comm = rasr_amd.Comm(ctx, 0, 1, uid)
    try:
        assert comm.rank == 0 and comm.world == 1
        model = synth.gmm_cart(200, 1, 16, 40, seed=61, pooled=True)
        sc = rasr_amd.GmmFeatureScorer(ctx, model)
        T, M = 3000, 200
        x = np.random.Generator(np.random.PCG64(62)).standard_normal((T, 40)).astype(np.float32)
        red = EpochReduceBuffer([("acc", sc.accumulator_size(), "f64"), ("score_sum", 1, "f64"), ("counts", M, "count")], device="cuda")
        xd = torch.from_numpy(x).cuda()
        scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
        bestd = torch.empty((T, M), dtype=torch.int32, device="cuda")
        state = torch.empty((T,), dtype=torch.int32, device="cuda")
        ctx.use_torch_stream()
        sc.score_stats_dev(xd, T, scores, bestd, state, red.view("counts"), red.view("score_sum"))
        sc.accumulate_dev(xd, T, state, bestd, M, red.view("acc"))
        red.view("counts")[3] += 2 ** 40   # beyond f32, within the exact range of an f64 slot
        torch.cuda.synchronize()
        before = (red.view("acc").cpu().numpy().copy(), red.view("counts").cpu().numpy().copy(), float(red.view("score_sum").item()))
        assert before[1].sum() == T + 2 ** 40 and before[0].sum() != 0
        calls = []
        real = comm.all_reduce_f64
        comm.all_reduce_f64 = lambda t: (calls.append(t.numel()), real(t))[1]
        red.all_reduce(comm=comm)
        torch.cuda.synchronize()
        assert calls == [sc.accumulator_size() + 1 + M]
        assert np.array_equal(red.view("acc").cpu().numpy(), before[0]) and np.array_equal(red.view("counts").cpu().numpy(), before[1])
        assert float(red.view("score_sum").item()) == before[2]
    finally:
        comm.close()
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.Comm(ctx, 2, 2, uid)   # rank out of range


# ------------------------------------------------- the PRODUCT branch of bench.py's main() at N = 2, device calls replaced in the test

def _run_fake_device_bench(argv, env_extra=None, timeout=400):
    """tests/fake_device_bench.py <argv>: bench.main() with every device call replaced by a host stand-in INSIDE that test file
    (bench.py and rasr_amd/ have no such switch).  Returns (rc, every JSON line of rank 0, stderr)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "fake_device_bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=root)
    return p.returncode, [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")], p.stderr


@pytest.mark.parametrize("workload", ["pipeline", "nn-pipeline"])
def test_bench_product_control_flow_at_two_ranks(workload):
    """The command the driver will run on an 8-GPU node, at N = 2 and without a GPU: `bench.py --gpus 2 --steps K --warmup W` --
    bench.py's OWN launcher starts the ranks (torch.distributed.run, 127.0.0.1), every rank goes through the GPU branch of main():
    Context, compute stream, make_comm (unique id broadcast over the control plane), the Pipeline job with the STREAMED ingest of its
    corpus partition, W + K steps, ONE amx_comm all-reduce of the flat EpochReduceBuffer, MAX of the times over the ranks, rank 0's JSON
    line.  Device calls are stand-ins that count frames (tests/fake_device_bench.py); what is checked is the plumbing: n_gpus, the
    weak-scaling unit count, the partition sizes, that the reduce summed BOTH ranks' frames and that it was one collective."""
    K, W, U = 3, 1, 4
    rc, lines, err = _run_fake_device_bench(["--gpus", "2", "--steps", str(K), "--warmup", str(W), "--utterances", str(U), "--utt-seconds", "1",
                                             "--workload", workload])
    assert rc == 0, err[-3000:]
    line = next(l for l in lines if "metric" in l)
    fake = next(l for l in lines if l.get("fake_device"))
    assert line["build"].startswith("fake device")          # the stand-ins ran, not a GPU
    assert line["n_gpus"] == 2 and line["steps"] == K and line["warmup"] == W and line["scaling"] == "weak" and line["higher_is_better"]
    F = U * 99                                               # 1 s at 16 kHz: 99 frames per utterance (Signal/WindowBuffer.cc)
    assert line["config"]["frames_per_step_per_gpu"] == F
    assert abs(line["value"] - 2 * F * K / (line["ms_per_step"] * 1e-3 * K)) <= 1e-3 * line["value"]   # whole-job units over the MAX time
    assert line["ingest"]["mode"] == "streamed" and line["ingest"]["sample_format"] == "s16"
    assert line["ingest"]["rank_partition_utterances"] * 2 == line["ingest"]["corpus_utterances"]       # i % 2 == rank
    er = line["epoch_reduce"]
    assert er["collectives"] == 1 and "2 ranks" in er["backend"]
    assert er["reduce_ok"] and er["reduced_frames"] == er["expected_frames"] == 2 * F * (K + W)
    assert fake["all_reduce_f64_calls_on_rank0"] == 1        # literally one collective per epoch
    assert "cpu_baseline" not in line and "configs" not in line   # rank 0 at N > 1 reports the job, not the 1-GPU extras


def test_bench_product_branch_refuses_a_world_size_mismatch():
    rc, lines, err = _run_fake_device_bench(["--gpus", "4", "--steps", "1", "--warmup", "0", "--utterances", "2", "--utt-seconds", "1"],
                                            env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "refusing to report a wrong n_gpus" in err
