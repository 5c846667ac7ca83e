#!/usr/bin/env python3
"""bench.py -- frames scored per second on MI355X (metric of BASELINE.json).

Default workload ("pipeline"): one step = one batch of synthetic 16 kHz utterances, resident in HBM, through the
whole hot path: fused MFCC-40 kernel -> 11-frame context window -> FFNN emission scorer 440 -> 6 x 2048 (ReLU)
-> 10 000 states (bf16 MFMA) -> per-frame best state + per-state counts (the epoch accumulators).  A frame counts
as "scored" when its 10 000 emission scores exist in HBM.  Other workloads time one stage on its BASELINE config:
  --workload mfcc      config 2  (batched MFCC-40, HBM roofline)
  --workload gmm       config 3  (CART-style instance: 10 000 states x 16 densities, pooled covariance, batch 256)
  --workload gmm-tied  config 3  (tied-mixture instance: 4096 shared densities, 10 000 states)
  --workload nn        config 4  (6x2048 FFNN, batch 1024)
  --workload gmm-train config 5, GMM leg: MFCC -> 10 000 x 16 GMM scoring -> Viterbi statistics (f64 weights, sum x, sum x^2;
                       a 104 MB accumulator) -> ONE all-reduce of the accumulator per epoch

Launch: `python bench.py` (1 GPU), `python bench.py --gpus N` (starts its own N ranks through torch.distributed.run on
127.0.0.1) or `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (ranks already started: RANK /
LOCAL_RANK / WORLD_SIZE in the environment; WORLD_SIZE must equal --gpus).  Utterances shard across ranks by the
reference's partition rule (weak scaling: every rank owns a full batch); the only data-path collective is ONE all-reduce of
the accumulators at the end of the timed epoch, through the library's own entry point (amx_comm_all_reduce_f64_dev = RCCL);
torch.distributed carries the barrier, the max over ranks of the time and the 128-byte communicator id.  Rank 0 prints one
JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak
FP32_TFLOPS = 157.3        # f32 vector (= f32 MFMA) peak


TRAFFIC_SOURCE = os.path.join("profiles", "r05", "traffic.json")


def measured_traffic(workload, *needles):
    """HBM bytes per launch measured OFFLINE in separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the `workload` bench
    (TRAFFIC_SOURCE, keys "workload | kernel name"; FETCH doubled for gfx950 by tools/traffic_json.py), or None when no
    entry of that workload names every needle (a different launch shape than the profiled one gets None, not a stale figure).
    It is a constant from the profiling box, not a measurement of this run: the line says so in `traffic_source`."""
    try:
        t = json.load(open(os.path.join(ROOT, TRAFFIC_SOURCE)))
        for k, v in t.items():
            w, _, name = k.partition(" | ")
            if w == workload and all(n in name for n in needles):
                return v["traffic"]
    except Exception:
        pass
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "nn-pipeline", "mfcc", "gmm", "gmm-tied", "nn", "gmm-train", "gmm-trained", "null"],
                    help="null: host-only stand-in (no GPU, --backend gloo): launcher, rendezvous, partitioning, epoch reduce and the JSON line")
    ap.add_argument("--backend", default=os.environ.get("AMX_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch.distributed backend of the control plane: nccl (= RCCL; GPU runs) or gloo (the CPU launcher test)")
    ap.add_argument("--utterances", type=int, default=64, help="utterances per step and rank (pipeline / mfcc)")
    ap.add_argument("--utt-seconds", type=float, default=10.0)
    ap.add_argument("--ingest", default="auto", choices=["auto", "resident", "streamed"],
                    help="pipeline / nn-pipeline / gmm-train: where a step's audio comes from.  resident: the same 64 utterances, f32, already in "
                         "HBM (the single-GPU `value`: inputs resident when the timed region starts).  streamed: every step takes the NEXT "
                         "utterances of the rank's corpus partition as s16 from pinned host memory over the host link into a "
                         "double-buffered HBM slot (copy stream, events), so the timed region carries the ingest.  auto = resident on "
                         "one GPU (the line then also reports the streamed rate next to it), streamed for --gpus N > 1")
    ap.add_argument("--corpus-hours", type=float, default=100.0, help="size of the synthetic corpus all ranks share (config 5: 100 h)")
    ap.add_argument("--trained-frames-per-state", type=int, default=600, help="gmm-trained: training frames per state of the split-trained model")
    ap.add_argument("--best-density", choices=["u8", "u32", "aligned"], default="u32",
                    help="best densities of the GMM leg: u32 = the matrix [frames x 10000] (Mm::DensityInMixture as RASR declares it); u8 = the "
                         "same matrix in bytes (amx_gmm_score_stats_u8_dev: a quarter of the memory, the same time, profiles/r04/gmm_store_ab.log); "
                         "aligned = no matrix: every state scored without index bookkeeping, then amx_gmm_best_density_dev for the aligned "
                         "(here: best) state of each frame -- what AssigningContextScorer::bestDensity(e) is asked for in Viterbi accumulation")
    ap.add_argument("--contract", default="fma", choices=["fma", "off"],
                    help="which build of the reference the GMM scorers are bit-identical to (amx_gmm_model.tuning contract=...): fma = its DEFAULT "
                         "configuration, -march=native with GCC's -ffp-contract=fast on an FMA host (the distance accumulates with fused "
                         "multiply-adds); off = configured with -DMARCH=x86-64.  The oracle that checks the run and the CPU baseline are built "
                         "the same way")
    ap.add_argument("--gmm-tuning", default=None, help='amx_gmm_model.tuning of every GMM scorer the workload builds, e.g. "screen=0" (A/B runs)')
    ap.add_argument("--nn-tuning", default=None, help='amx_ffnn_model.tuning, e.g. "tile=4" or "graph=1"')
    ap.add_argument("--mfcc-tuning", default=None, help='amx_mfcc_cfg.tuning, e.g. "fft=mfma" or "wgs=3"')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary BASELINE configs of the default run")
    ap.add_argument("--precision", default="f16mx", choices=["bf16", "bf16x3", "f16mx", "fp32"],
                    help="NN GEMM inputs: f16mx = f16 product + one MX-fp6 scaled product for both cross terms (default since round 4: meets "
                         "north_star's 1e-4 bar against the f32 reference at 1.5 units of matrix time per product), bf16x3 = split bf16, "
                         "three MFMA products per f32 product (the round-3 default, same bar), bf16 (BASELINE config 4's literal dtype; "
                         "2e-3-grade scores), fp32 = f32 MFMA")
    ap.add_argument("--front-end", default="mfcc", choices=["mfcc", "mfplp", "plp", "gammatone"],
                    help="mfcc workload: mfcc.flow (40 cepstra), mfplp.flow (20 autocorrelation / 16 cepstrum coefficients) or plp.flow "
                         "(bark / trapeze filter bank + equal loudness, 13 / 13) or the gammatone nodes (68 channels, cascade 4, 25 / 10 ms "
                         "Hanning temporal integration, spectral integration 9 / 4, 10th root, 12 cepstra)")
    ap.add_argument("--estimation-mode", default="viterbi", choices=["viterbi", "baum-welch"],
                    help="gmm-train: statistics of the best density only, or of every density by its posterior (reference: mode)")
    ap.add_argument("--gmm-type", default="diagonal-maximum", choices=["diagonal-maximum", "batch-diagonal-maximum-float", "SIMD-diagonal-maximum"])
    ap.add_argument("--gmm-frames", type=int, default=256, help="gmm / gmm-tied: frames per step (config 3: 256)")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (torch.distributed.run, rendezvous on 127.0.0.1) and
    hand their exit code on.  Returns only when this process IS a rank (or N = 1)."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d: refusing to report a wrong n_gpus" % (world, args.gpus))
        return
    if args.gpus <= 1:
        return
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC between the ranks' GPUs (RCCL)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    sys.exit(subprocess.call(cmd, env=env))


def dist_setup(args):
    """(rank, world, local rank).  One process per GPU; the process group is the control plane only."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = args.backend == "nccl"
    if gpu:
        import torch
        if local >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d wants GPU %d but only %d are visible" % (rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
    if world > 1 or os.environ.get("AMX_BENCH_FORCE_DIST"):  # the env switch runs the RCCL path with a single rank (1-GPU boxes)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    return rank, world, local


def make_comm(ctx, rank, world):
    """the data-path communicator (amx_comm_*): rank 0's id travels through the control-plane group"""
    import torch
    import torch.distributed as dist

    import rasr_amd
    if not _dist_on():
        return None
    uid = [rasr_amd.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = rasr_amd.Comm(ctx, rank, world, uid[0])
    if comm.world != dist.get_world_size():
        raise SystemExit("bench.py: RCCL communicator has %d ranks, the process group %d" % (comm.world, dist.get_world_size()))
    torch.cuda.synchronize()
    return comm


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def barrier(world, gpu=True):
    if _dist_on():
        import torch.distributed as dist
        dist.barrier()
    if gpu:
        import torch
        torch.cuda.synchronize()


def make_batch(n_utt, seconds, seed):
    """n_utt utterances of `seconds` s: one synthetic waveform, rotated per utterance (cheap, all different)."""
    from tests import synth
    n = int(round(seconds * 16000))
    base = synth.waveform(n + n_utt, seed=seed)
    pcm = np.concatenate([base[u:u + n] for u in range(n_utt)])
    off = np.arange(n_utt + 1, dtype=np.int64) * n
    return pcm, off


def ingest_mode(args, world):
    m = getattr(args, "ingest", "auto")
    return m if m != "auto" else ("streamed" if world > 1 else "resident")


class StreamedIngest:
    """The audio of BASELINE config 5 as a rank of the job sees it: a corpus of `--corpus-hours` of utterances, of which this
    rank owns the partition `i % world == rank` (rasr_amd.partition.CorpusWalker = the reference's partition rule,
    Bliss/CorpusDescription.cc:174-190).  The samples wait in pinned host memory as s16 -- what the audio files hold; the kernel
    widens them like Flow/TypeConverter.hh:35-43 -- and every step's batch crosses the host link inside the timed region:
    one hipMemcpyAsync per batch on a copy stream into one of two HBM slots; the MFCC kernel of step k waits for slot k % 2's
    event, the copy of batch k + 2 waits for the event recorded behind that kernel.

    Host memory holds the rank's first `resident_batches` batches (a window of the partition: generating 12.5 h of synthetic
    audio per rank would only lengthen start-up); the walker keeps handing out the partition's real indices and the window
    wraps.  All utterances have `seconds` s, so one MFCC plan serves every batch."""

    def __init__(self, args, rank, world, n_steps):
        import torch

        from rasr_amd.partition import CorpusWalker
        from tests import synth
        self.torch = torch
        n = int(round(args.utt_seconds * 16000))
        self.n, self.B = n, args.utterances
        n_global = max(int(args.corpus_hours * 3600.0 / args.utt_seconds), self.B * world)
        self.walker = CorpusWalker(n_global, world, rank, self.B)
        self.n_global, self.n_local = n_global, len(self.walker.segments)
        nb = max(2, min(self.walker.batches_per_epoch(), n_steps + 2))
        base = synth.waveform(n + 8192, seed=9).astype(np.int16)       # utterance g = the base waveform rotated by g (all different)
        host = torch.empty((nb * self.B, n), dtype=torch.int16).pin_memory()
        hv = host.numpy()
        w = CorpusWalker(n_global, world, rank, self.B)
        self.window = []
        for b in range(nb):
            ids = w.next_batch()
            self.window.append(ids)
            for j, g in enumerate(ids):
                r = int(g) % 8192
                hv[b * self.B + j] = base[r:r + n]
            for j in range(len(ids), self.B):                          # short last batch of the partition: padded with silence
                hv[b * self.B + j] = 0
        self.host, self.nb = host, nb
        self.slots = [torch.empty((self.B * n,), dtype=torch.int16, device="cuda") for _ in range(2)]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [torch.cuda.Event(), torch.cuda.Event()]
        self.copy_stream = torch.cuda.Stream()
        self.issued = 0          # batches handed to the copy stream
        self.taken = 0           # batches handed to the compute stream
        self.visited = []        # corpus indices, in the order this rank scored them
        self.bytes_per_batch = self.B * n * 2
        self._issue(first=True)
        self._issue(first=True)

    def _issue(self, first=False):
        torch = self.torch
        slot = self.issued % 2
        b = self.issued % self.nb
        with torch.cuda.stream(self.copy_stream):
            if not first:
                self.copy_stream.wait_event(self.free[slot])           # the kernel that read this slot has finished
            self.slots[slot].view(self.B, self.n).copy_(self.host[b * self.B:(b + 1) * self.B], non_blocking=True)
            self.ready[slot].record(self.copy_stream)
        self.issued += 1

    def take(self):
        """the next batch, as an s16 tensor in HBM; the caller's stream waits for its copy"""
        slot = self.taken % 2
        self.torch.cuda.current_stream().wait_event(self.ready[slot])
        self.visited.append(self.walker.next_batch())
        return self.slots[slot]

    def release(self):
        """behind the kernel that read the batch: the slot may be overwritten, the next copy starts"""
        slot = self.taken % 2
        self.free[slot].record(self.torch.cuda.current_stream())
        self.taken += 1
        self._issue()

    def report(self):
        return dict(mode="streamed", sample_format="s16", bytes_over_link_per_step=self.bytes_per_batch, corpus_utterances=self.n_global,
                    rank_partition_utterances=self.n_local, host_window_batches=self.nb,
                    path="pinned host s16 -> hipMemcpyAsync (copy stream) -> 2 HBM slots -> amx_mfcc_run_plan_dev_s16 behind an event")


def gmm_cart_roofline(ctx, sc, nk, n_mix, dim, frames, best_bytes=4):
    """roofline entry of the screened private-density GMM scorer.

    Fused path (gmm_fused_kernel): `achieved` is the f32 arithmetic the kernel really issues for the reference's distance --
    (densities evaluated exactly, from a device counter) x 4 dim flops (contract=off: sub, mul, mul, add -- unfused by definition of
    the reference's -DMARCH=x86-64 build; contract=fma: sub, mul, fma -- the same four flops in three instructions, the reference's
    default build) -- over the kernel's HIP-event time, priced against the f32 vector peak; `frac` is therefore a
    utilisation <= 1.  The reference scorer's algorithmic flops (every density) over the same time are reported
    separately as `algorithmic_speedup_vs_dense` (how much faster than a dense f32 evaluation at peak), and the HBM side as
    `hbm_algorithmic_GBps` (scores + best densities out, features + model records in)."""
    ms_x, n_x = ctx.profile_get("gmm")
    ms_s, n_s = ctx.profile_get("gmm_screen")
    ms_p, n_p = ctx.profile_get("gmm_screen_pack")
    if n_x == 0:
        return None
    surv, pairs = sc.screen_counts(False)
    alg = (3.0 * dim + 2.0) * nk * frames   # SURVEY 8(d) cfg 3 secondary: D (3d + 2) flop per frame for the reference scorer
    if pairs:  # fused kernel: one launch does screen + exact evaluation
        t = ms_x * 1e-3
        per_launch = surv / float(n_x)
        ex = per_launch * 4.0 * dim
        by = frames * (n_mix * (4.0 + best_bytes) + dim * 4.0) + (n_mix + 15) // 16 * 78848.0
        scr = 2.0 * (16 * ((dim + 2 + 15) // 16)) * ((n_mix + 15) // 16 * 256) * frames   # K-steps that hold non-zero columns (dim + 2)
        return dict(bound="valu", kernel="gmm_fused_kernel<%d> (f16 MFMA screen + exact f32/f64 evaluation of the survivors)" % dim,
                    note="VALU-issue-bound kernel priced against the f32 vector peak (157.3 TFLOP/s; unfused mul/add can "
                         "reach half of it). achieved = densities evaluated exactly (device counter) x 4 dim f32 flops (sub, mul, mul, add; "
                         "contract=fma: sub, mul, fma) / kernel time",
                    achieved=round(ex / t / 1e12, 2), peak=FP32_TFLOPS, unit="TFLOP/s", frac=round(ex / t / 1e12 / FP32_TFLOPS, 4),
                    traffic=measured_traffic("pipeline", "gmm_fused_kernel", "Li%dELi%d" % (dim, {0: 0, 1: 2}.get(best_bytes, 1)))
                    if (n_mix == 10000 and frames == 63936) else None,
                    best_density_bytes=best_bytes,
                    avg_launch_ms=round(ms_x, 4), pack_launch_ms=round(ms_p, 4), launches=n_x, flops_per_launch=ex,
                    survivors_per_mixture=round(surv / float(pairs), 4),
                    screen_mfma_tflops=round(scr / t / 1e12, 1), screen_mfma_frac=round(scr / t / 1e12 / MFMA_BF16_TFLOPS, 4),
                    hbm_algorithmic_GBps=round(by / t / 1e9, 1), hbm_frac=round(by / t / 1e9 / HBM_PEAK_GBS, 4),
                    algorithmic_speedup_vs_dense=round(alg / (t + ms_p * 1e-3) / 1e12 / FP32_TFLOPS, 3))
    if n_s == 0:  # screen disabled (--gmm-tuning screen=0): the exact-everything kernel
        ops = 4.0 * nk * dim * frames
        ach = ops / (ms_x * 1e-3) / 1e12
        return dict(bound="valu", note="f32 VALU kernel priced against the f32 vector peak", kernel="gmm_direct_kernel<%d,MaxState>" % dim,
                    achieved=round(ach, 3), peak=FP32_TFLOPS, unit="TFLOP/s", frac=round(ach / FP32_TFLOPS, 4), traffic=None,
                    avg_launch_ms=round(ms_x, 4), launches=n_x, flops_per_launch=ops)
    # two-kernel path (--gmm-tuning fused=0, per-density covariances, dim > 40): HBM-side figure of the exact stage (scores, best
    # densities, survivor masks) -- no survivor counter there
    t = ms_x * 1e-3
    by = frames * (n_mix * 8.0 + ((n_mix + 15) // 16 * 16) * 2.0 + dim * 4.0)
    return dict(bound="hbm", kernel="gmm_screen_exact_kernel<%d> (+ gmm_screen_rows_kernel, gmm_screen_pack_kernel)" % dim,
                achieved=round(by / t / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(by / t / 1e9 / HBM_PEAK_GBS, 4),
                traffic=None,
                avg_launch_ms=round(ms_x, 4), screen_launch_ms=round(ms_s, 4), pack_launch_ms=round(ms_p, 4), launches=n_x,
                bytes_per_launch=by, algorithmic_speedup_vs_dense=round(alg / ((ms_x + ms_s + ms_p) * 1e-3) / 1e12 / FP32_TFLOPS, 3))


def nn_gemm_roofline(precision, ms, n, frames, full_chunk):
    """output-layer GEMM 2048 -> 10000: `achieved` = MFMA flops the kernel executes / HIP-event time.  bf16x3 executes three bf16
    products per algorithmic product, so its algorithmic rate is a third of `achieved`."""
    alg = 2.0 * 2048 * 10000 * frames
    # f16mx: the f16 product (1 x alg flops at the 2.5 PF rate) + the fp6 x fp6 scaled product (2 x alg flops at the 10 PF rate) keep
    # the matrix pipe busy like 1.5 x alg flops at the 2.5 PF rate: `achieved` is that bf16-equivalent figure, so `frac` is pipe time
    mult = {"bf16x3": 3.0, "f16mx": 1.5}.get(precision, 1.0)
    peak = FP32_TFLOPS if precision == "fp32" else MFMA_BF16_TFLOPS
    ach = mult * alg / (ms * 1e-3) / 1e12
    name = {"bf16": "gemm_bf16_pipe_kernel<NONE,LAST> (2048->10000)", "bf16x3": "gemm_bf16_pipe_kernel<X3,NONE,LAST> (2048->10000, split bf16: W_hi/W_lo/X_hi/X_lo staged once, 3 MFMA products per fragment set)",
            "f16mx": "gemm_mx_kernel<256x256,NONE,LAST> (2048->10000; per 32 k: 2 x v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 "
                     "fp6 x fp6 = 1.5 f16-equivalent units per product)",
            "fp32": "gemm_f32_kernel (2048->10000)"}[precision]
    out = dict(bound="mfma", kernel=name, achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
               traffic=(measured_traffic({"bf16x3": "pipeline-bf16x3", "bf16": "pipeline-bf16", "f16mx": "pipeline"}[precision],
                                         "gemm_mx_kernel" if precision == "f16mx" else "gemm_bf16_pipe_kernel",
                                         {"bf16x3": "32, true>, 0, true", "bf16": "64, false>, 0, true", "f16mx": ">, 0, true, 0>"}[precision])
                        if (precision in ("bf16", "bf16x3", "f16mx") and full_chunk) else None),
               avg_launch_ms=round(ms, 4), launches=n, flops_per_launch=mult * alg)
    if mult != 1.0:
        out["algorithmic_tflops"] = round(alg / (ms * 1e-3) / 1e12, 2)
    return out


class NnPipeline:
    """MFCC-40 -> context 11 -> FFNN 440-6x2048-10000 -> accumulators, everything resident in HBM."""

    CHUNK = int(os.environ.get("AMX_BENCH_CHUNK", "32768"))  # frames per scoring pass (bounds the [frames x 10000] f32 score buffer to 1.3 GB)

    def __init__(self, ctx, args, rank):
        import torch

        import rasr_amd
        from tests import synth
        self.torch, self.ctx = torch, ctx
        self.fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0, tuning=args.mfcc_tuning)
        pcm, off = make_batch(args.utterances, args.utt_seconds, seed=9 + rank)
        self.plan = self.fe.plan(off)
        self.F = self.plan.total_frames
        self.pcm = torch.from_numpy(pcm).cuda()
        self.setup_ingest(args, rank)
        self.ceps = torch.empty((self.F, 40), dtype=torch.float32, device="cuda")
        self.ctxwin = torch.empty((self.F, 440), dtype=torch.float32, device="cuda")
        dims = [440] + [2048] * 6 + [10000]
        Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
        self.nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision=args.precision, tuning=args.nn_tuning)
        self.flops_per_frame = 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
        self.M = 10000
        self.scores = torch.empty((min(self.CHUNK, self.F), self.M), dtype=torch.float32, device="cuda")
        self.best = torch.empty((self.F,), dtype=torch.int32, device="cuda")
        self.make_reduce_buffer()
        self.units = self.F

    def setup_ingest(self, args, rank):
        self.args, self.rank, self.world = args, rank, getattr(args, "_world", 1)
        self.ingest = StreamedIngest(args, rank, self.world, args.steps + args.warmup) if ingest_mode(args, self.world) == "streamed" else None

    def torch_mod(self):
        import torch
        return torch

    def front_end(self):
        """audio -> cepstra: from the resident batch, or from the next batch of the rank's corpus partition (StreamedIngest)"""
        if self.ingest is None:
            self.fe.run_plan(self.plan, self.pcm, self.ceps)
            return
        pcm = self.ingest.take()
        self.fe.run_plan(self.plan, pcm, self.ceps)
        self.ingest.release()

    def reduce_fields(self):
        return [("score_sum", 1, "f64"), ("counts", self.M, "count")]

    def make_reduce_buffer(self):
        """all cross-rank state of the job in ONE flat buffer: epoch_reduce is one all-reduce"""
        from rasr_amd.partition import EpochReduceBuffer
        self.red = EpochReduceBuffer(self.reduce_fields(), device="cuda")
        self.counts = self.red.view("counts")
        self.score_sum = self.red.view("score_sum")

    def step(self):
        self.front_end()
        self.ctx.context_window(self.plan, self.ceps, 40, 5, 5, self.ctxwin, 440)
        for t0 in range(0, self.F, self.CHUNK):
            T = min(self.CHUNK, self.F - t0)
            self.nn.score_stats_dev(self.ctxwin[t0:], 440, T, self.scores, self.best[t0:], self.counts, self.score_sum)

    def epoch_reduce(self, world):
        self.red.all_reduce(comm=getattr(self, "comm", None))   # ONE collective (amx_comm_all_reduce_f64_dev; no-op in a single process)

    def roofline(self):
        """The GEMM TEMPLATE of the NN leg (the kernel with the largest summed time of the step): algorithmic flops of all seven layers
        (84.7 MFLOP per frame) over the summed HIP-event time of their launches, against the dense bf16 / f16 matrix peak.
        `frac` is ALGORITHMIC (one f32 product of the reference = one unit); what the matrix pipes execute for it is next to it:
        f16mx = one f16 product + half an fp6 x fp6 32x32x64 product per 16 k, nominally 1.5 units -- measured, the fp6 product costs
        1.6 f16 products on this chip (tools/feed_probe.hip), i.e. 1.8 units; bf16x3 = 3 units.  The output layer alone: `largest_launch`."""
        ms_o, n_o = self.ctx.profile_get("ffnn_gemm_max")
        ms_h, n_h = self.ctx.profile_get("ffnn_gemm")
        if n_o == 0:
            return None
        rows = []
        for t0 in range(0, self.F, self.CHUNK):
            rows.append(min(self.CHUNK, self.F - t0))
        frames = sum(rows) / len(rows)
        largest = nn_gemm_roofline(self.nn_precision, ms_o, n_o, frames, self.F >= self.CHUNK)
        t_ms = ms_h * n_h                                     # every GEMM launch of the profiled steps ("ffnn_gemm" times all seven layers of a
                                                              # pass, "ffnn_gemm_max" the output layer once more on its own)
        alg = self.flops_per_frame * frames * n_o             # one output-layer launch per scoring pass
        peak = FP32_TFLOPS if self.nn_precision == "fp32" else MFMA_BF16_TFLOPS
        ach = alg / (t_ms * 1e-3) / 1e12
        units = {"bf16x3": (3.0, 3.0), "f16mx": (1.5, 1.8)}.get(self.nn_precision, (1.0, 1.0))
        return dict(bound="mfma", kernel={"f16mx": "gemm_mx_kernel", "fp32": "gemm_f32_kernel"}.get(self.nn_precision, "gemm_bf16_pipe_kernel") +
                    " (all 7 GEMMs of the forward pass: 440-6x2048-10000; the template with the largest summed time of the step)",
                    achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                    note="ALGORITHMIC: 2 x 42.35 M products per frame (SURVEY 8(d)) over the summed time of the %d GEMM launches" % n_h,
                    executed_units_per_product=dict(nominal=units[0], measured_on_this_chip=units[1],
                                                    frac_of_peak_in_executed_units=round(units[1] * ach / peak, 4),
                                                    what="f16mx: 1 f16 MFMA product + 0.5 fp6 x fp6 scaled product per 16 k; the scaled product "
                                                         "costs 1.6 f16 products here (feed probe), nominally 1" if self.nn_precision == "f16mx" else
                                                         "matrix products executed per f32 product of the reference"),
                    traffic=self.template_traffic(largest), avg_launch_ms=round(t_ms / n_h, 4), launches=n_h, flops_per_launch=alg / n_h,
                    summed_ms_per_step=round(t_ms / max(1, n_o // len(rows)), 4), largest_launch=largest)

    def template_traffic(self, largest):
        """HBM-side bytes per launch of the GEMM template, averaged over the seven launches of a pass like `achieved`: six hidden-layer
        launches (one PMC entry: same instantiation) and the output layer's, from the offline FETCH_SIZE / WRITE_SIZE passes"""
        if self.nn_precision != "f16mx" or self.F < self.CHUNK or largest.get("traffic") is None:
            return None
        hid = measured_traffic("pipeline", "gemm_mx_kernel", ">, 1, false, 0>")
        return None if hid is None else round((6.0 * hid + largest["traffic"]) / 7.0, 1)

    def stage_report(self):
        out = {}
        for k in ("mfcc", "context_window", "ffnn_pack", "ffnn_gemm", "ffnn_gemm_max", "ffnn_split", "stats"):
            ms, n = self.ctx.profile_get(k)
            if n:
                out[k] = dict(avg_ms=round(ms, 4), launches=n)
        ms, n = self.ctx.profile_get("mfcc")
        if n:
            gbs = self.F * 800.0 / (ms * 1e-3) / 1e9
            out["mfcc"].update(algorithmic_GBps=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK_GBS, 4))
        return out


class Pipeline(NnPipeline):
    """Per-GPU shard of BASELINE config 5: every frame is scored by BOTH acoustic models.
    audio -> MFCC-40 -+-> CART GMM 10 000 states x 16 densities (diagonal-maximum) -> best state / density -> Viterbi accumulators
                      +-> 11-frame context -> FFNN 440-6x2048-10000 (bf16 MFMA)   -> best state -> per-state counts
    The two legs only share the MFCC output (AMX_BENCH_TWO_STREAMS=1 runs them on two HIP streams: 1 % faster, per-kernel times no longer exclusive)."""

    GCHUNK = int(os.environ.get("AMX_BENCH_GCHUNK", "65536"))  # frames per GMM pass (>= a step: one pass; scores + best densities = 5.1 GB)

    def reduce_fields(self):
        return [("acc", self.gmm.accumulator_size(), "f64"), ("score_sum", 1, "f64"), ("gscore_sum", 1, "f64"),
                ("counts", self.M, "count"), ("gcounts", self.M, "count")]

    def __init__(self, ctx, args, rank):
        import torch

        import rasr_amd
        from tests import synth
        model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
        self.nk = int(model["mix_offsets"][-1])
        self.gmm = rasr_amd.GmmFeatureScorer(ctx, model, tuning=args.gmm_tuning)
        super().__init__(ctx, args, rank)
        g = min(self.GCHUNK, self.F)
        self.gscores = torch.empty((g, self.M), dtype=torch.float32, device="cuda")
        self.aligned = args.best_density == "aligned"
        self.gbestd = (torch.empty((g,), dtype=torch.int32, device="cuda") if self.aligned else
                       torch.empty((g, self.M), dtype=torch.uint8 if args.best_density == "u8" else torch.int32, device="cuda"))
        self.gstate = torch.empty((g,), dtype=torch.int32, device="cuda")
        self.gcounts = self.red.view("gcounts")
        self.gscore_sum = self.red.view("gscore_sum")
        self.acc = self.red.view("acc")   # the 53.8 MB of f64 GMM statistics accumulate in place in the reduce buffer
        self.main = torch.cuda.current_stream()
        self.side = torch.cuda.Stream()
        self.ev_feat, self.ev_side = torch.cuda.Event(), torch.cuda.Event()

    def gmm_leg(self):
        for t0 in range(0, self.F, self.GCHUNK):
            T = min(self.GCHUNK, self.F - t0)
            x = self.ceps[t0:]
            if self.aligned:
                self.gmm.score_stats_dev(x, T, self.gscores, None, self.gstate, self.gcounts, self.gscore_sum)
                self.gmm.best_density_dev(x, T, self.gstate, self.gbestd)
                self.gmm.accumulate_dev(x, T, self.gstate, self.gbestd, 0, self.acc)
            else:
                self.gmm.score_stats_dev(x, T, self.gscores, self.gbestd, self.gstate, self.gcounts, self.gscore_sum)
                self.gmm.accumulate_dev(x, T, self.gstate, self.gbestd, self.M, self.acc)

    def nn_leg(self):
        self.ctx.context_window(self.plan, self.ceps, 40, 5, 5, self.ctxwin, 440)
        for t0 in range(0, self.F, self.CHUNK):
            T = min(self.CHUNK, self.F - t0)
            self.nn.score_stats_dev(self.ctxwin[t0:], 440, T, self.scores, self.best[t0:], self.counts, self.score_sum)

    def step(self):
        torch = self.torch
        self.front_end()
        # Both legs saturate the GPU.  On two streams the step gains 1 % (12.56 -> 12.44 ms, two alternations) -- and every per-launch time
        # (HIP events here, rocprofv3's durations alike) then includes the other leg's kernels: the GEMM's roofline fraction read 0.213
        # instead of 0.272 for the same work.  One stream, so that a kernel's time is its own.
        if not os.environ.get("AMX_BENCH_TWO_STREAMS"):
            self.gmm_leg()
            self.nn_leg()
            return
        self.ev_feat.record(self.main)
        self.side.wait_event(self.ev_feat)
        with torch.cuda.stream(self.side):   # GMM leg on the side stream
            self.ctx.use_torch_stream()
            self.gmm_leg()
            self.ev_side.record(self.side)
        self.ctx.use_torch_stream()          # back on the main stream: NN leg
        self.nn_leg()
        self.main.wait_event(self.ev_side)   # the next step's MFCC overwrites ceps

    def roofline(self):
        nn = super().roofline()
        gm = gmm_cart_roofline(self.ctx, self.gmm, self.nk, self.M, 40, min(self.GCHUNK, self.F), 0 if self.aligned else self.gbestd.element_size())
        if gm is None:
            return nn
        # "dominant kernel" = the kernel TEMPLATE with the larger summed time in the step: gmm_fused_kernel or the GEMM (all its launches)
        t_gmm = gm["avg_launch_ms"] * gm["launches"]
        t_nn = nn["avg_launch_ms"] * nn["launches"] if nn else 0.0
        first, second = (gm, nn) if t_gmm >= t_nn else (nn, gm)
        first["second"] = second
        return first

    def stage_report(self):
        out = super().stage_report()
        out["accumulator_bytes"] = int(self.acc.numel() * 8)
        for k in ("gmm_screen_pack", "gmm_screen", "gmm", "gmm_best_density", "gmm_accumulate"):
            ms, n = self.ctx.profile_get(k)
            if n:
                out[k] = dict(avg_ms=round(ms, 4), launches=n)
        return out


class GmmTrain:
    """config 5, GMM leg: audio -> MFCC-40 -> CART GMM (10 000 states x 16 densities) -> best state / density -> accumulators"""

    CHUNK = int(os.environ.get("AMX_BENCH_GCHUNK", "65536"))

    def __init__(self, ctx, args, rank):
        import torch

        import rasr_amd
        from tests import synth
        self.ctx = ctx
        self.fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0, tuning=args.mfcc_tuning)
        pcm, off = make_batch(args.utterances, args.utt_seconds, seed=9 + rank)
        self.plan = self.fe.plan(off)
        self.F = self.plan.total_frames
        self.pcm = torch.from_numpy(pcm).cuda()
        self.setup_ingest(args, rank)
        self.ceps = torch.empty((self.F, 40), dtype=torch.float32, device="cuda")
        model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
        self.nk = int(model["mix_offsets"][-1])
        self.sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=args.gmm_tuning)
        self.M = 10000
        g = min(self.CHUNK, self.F)
        self.scores = torch.empty((g, self.M), dtype=torch.float32, device="cuda")
        self.bestd = torch.empty((g, self.M), dtype=torch.uint8 if args.best_density == "u8" else torch.int32, device="cuda")
        self.state = torch.empty((g,), dtype=torch.int32, device="cuda")
        from rasr_amd.partition import EpochReduceBuffer
        self.red = EpochReduceBuffer([("acc", self.sc.accumulator_size(), "f64"), ("score_sum", 1, "f64"), ("counts", self.M, "count")], device="cuda")
        self.counts, self.score_sum, self.acc = self.red.view("counts"), self.red.view("score_sum"), self.red.view("acc")
        self.units = self.F
        self.baum_welch = getattr(args, "estimation_mode", "viterbi") == "baum-welch"

    setup_ingest = NnPipeline.setup_ingest
    front_end = NnPipeline.front_end
    torch_mod = NnPipeline.torch_mod

    def step(self):
        import rasr_amd
        self.front_end()
        for t0 in range(0, self.F, self.CHUNK):
            T = min(self.CHUNK, self.F - t0)
            x = self.ceps[t0:]
            self.sc.score_stats_dev(x, T, self.scores, self.bestd, self.state, self.counts, self.score_sum)
            if self.baum_welch:
                self.sc.accumulate_weighted_dev(rasr_amd.AMX_GMM_BAUM_WELCH, x, T, self.state, None, None, 0, self.acc)
            else:
                self.sc.accumulate_dev(x, T, self.state, self.bestd, self.M, self.acc)

    def epoch_reduce(self, world):
        self.red.all_reduce(comm=getattr(self, "comm", None))   # ONE collective: 8 * (sum K + n_mean * (1 + d) + n_cov * (1 + d)) bytes = 53.8 MB here (+ counts, score sum)

    def roofline(self):
        return gmm_cart_roofline(self.ctx, self.sc, self.nk, self.M, 40, min(self.CHUNK, self.F), self.bestd.element_size())

    def stage_report(self):
        out = {"accumulator_bytes": int(self.acc.numel() * 8)}
        for k in ("mfcc", "gmm_screen_pack", "gmm_screen", "gmm", "stats", "gmm_accumulate", "gmm_accumulate_weighted"):
            ms, n = self.ctx.profile_get(k)
            if n:
                out[k] = dict(avg_ms=round(ms, 4), launches=n)
        return out


class GmmTrained(GmmTrain):
    """The GMM leg of config 5 on a trained-SHAPED model instead of the random-init one: 10 000 states grown from one density to (up
    to) 16 by the repository's own loop -- accumulate, amx_gmm_estimate + split, Viterbi re-estimation (tests/trained_gmm.py;
    Mm/MixtureSetSplitter.cc:38-123, Mm/AbstractMixtureSetEstimator.cc:117-150,305-338) -- on synthetic clustered features, scored on
    63 936 frames of those features.  The densities of a mixture are close relatives here, so more of them survive the f16 screen of
    gmm_fused_kernel than of the random-init model's (survivors_per_mixture); the line reports that, the time per pass and the
    evaluate-everything kernel (tuning screen=0) on the same model and frames."""

    def __init__(self, ctx, args, rank):
        import torch

        import rasr_amd
        from rasr_amd.partition import EpochReduceBuffer
        from tests.trained_gmm import split_trained_gmm
        self.ctx = ctx
        t0 = time.perf_counter()
        model, x, align, self.history = split_trained_gmm(ctx, n_mix=10000, dim=40, frames_per_state=args.trained_frames_per_state,
                                                           rounds=4, iters=2, seed=11 + rank)
        self.train_seconds = time.perf_counter() - t0
        self.F = 63936
        self.x = x[:self.F].clone()
        self.align = align[:self.F].clone()
        del x, align
        torch.cuda.empty_cache()
        self.model = model
        self.nk = int(model["mix_offsets"][-1])
        self.kmax = int(np.diff(model["mix_offsets"]).max())
        self.sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=args.gmm_tuning)
        self.args_gmm_tuning = args.gmm_tuning
        self.M = 10000
        self.scores = torch.empty((self.F, self.M), dtype=torch.float32, device="cuda")
        self.bestd = torch.empty((self.F, self.M), dtype=torch.int32, device="cuda")
        self.state = torch.empty((self.F,), dtype=torch.int32, device="cuda")
        self.red = EpochReduceBuffer([("acc", self.sc.accumulator_size(), "f64"), ("score_sum", 1, "f64"), ("counts", self.M, "count")], device="cuda")
        self.counts, self.score_sum, self.acc = self.red.view("counts"), self.red.view("score_sum"), self.red.view("acc")
        self.units = self.F
        self.ingest = None

    def step(self):
        self.sc.score_stats_dev(self.x, self.F, self.scores, self.bestd, self.state, self.counts, self.score_sum)
        self.sc.accumulate_dev(self.x, self.F, self.state, self.bestd, self.M, self.acc)

    def stage_report(self):
        import torch

        import rasr_amd
        out = super().stage_report()
        out["model"] = dict(densities=self.nk, max_densities_per_mixture=self.kmax, training=self.history, training_seconds=round(self.train_seconds, 1),
                            best_state_is_aligned_state=round(float((self.state == self.align).float().mean()), 4))
        # the evaluate-everything kernel on the same model and frames (what a scorer without the screen costs)
        dense = rasr_amd.GmmFeatureScorer(self.ctx, self.model, tuning="screen=0")
        s2, b2 = torch.empty_like(self.scores), torch.empty_like(self.bestd)
        dense.score_dev(self.x, self.F, s2, b2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dense.score_dev(self.x, self.F, s2, b2)
        torch.cuda.synchronize()
        out["dense_kernel_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
        self.sc.score_dev(self.x, self.F, self.scores, self.bestd)
        torch.cuda.synchronize()
        out["screened_equals_dense_bitwise"] = bool(torch.equal(s2.view(torch.int32), self.scores.view(torch.int32)) and torch.equal(b2, self.bestd))
        # the worst case for the screen: the model as the trainer writes it right after its last split (every mean with >= 20
        # observations replaced by the twins mean +- eps sqrt(var), Mm/MixtureSetSplitter.cc:59-75), before any re-estimation
        from tests.trained_gmm import split_trained_gmm
        twins = getattr(split_trained_gmm, "fresh_split", None)
        if twins is not None:
            tw = rasr_amd.GmmFeatureScorer(self.ctx, twins, tuning=self.args_gmm_tuning)
            tw.score_dev(self.x, self.F, s2, b2)
            tw.screen_counts(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                tw.score_dev(self.x, self.F, s2, b2)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 3
            surv, pairs = tw.screen_counts(False)
            out["just_split_twins"] = dict(densities=int(twins["mix_offsets"][-1]), ms_per_pass=round(ms, 3),
                                           survivors_per_mixture=round(surv / float(max(pairs, 1)), 4),
                                           note="exact twins survive the f16 screen together; the next re-estimation separates them")
        return out


class MfccOnly:
    def __init__(self, ctx, args, rank):
        import torch

        import rasr_amd
        from tests import synth
        self.ctx = ctx
        fe = getattr(args, "front_end", "mfcc")
        self.plp = fe != "mfcc"
        self.nout = {"mfcc": 40, "mfplp": 16, "plp": 13, "gammatone": 12}[fe]
        self.gt = fe == "gammatone"
        if self.gt:
            self.fe = rasr_amd.GammatoneExtractor(ctx, channels=68, max_freq=7500.0, si_length=9, si_shift=4, power=0.1, n_ceps=12)
        elif fe == "plp":
            self.fe = rasr_amd.MfccExtractor.plp(ctx, tuning=args.mfcc_tuning)
        elif self.plp:
            self.fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=16, front_end="mfplp", nr_autocorrelation_coefficients=20,
                                             normalize=True, tuning=args.mfcc_tuning)
        else:
            self.fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0, tuning=args.mfcc_tuning)
        lens = synth.utterance_lengths(1000, seed=3)
        base = synth.waveform(int(lens.max()) + 1000, seed=4 + rank)
        pcm = np.concatenate([base[u:u + int(n)] for u, n in enumerate(lens)])
        off = np.concatenate([[0], np.cumsum(lens)])
        if self.gt:
            self.off = off.astype(np.int64)
            self.F = int(sum(self.fe.n_frames(int(n)) for n in lens))
        else:
            self.plan = self.fe.plan(off)
            self.F = self.plan.total_frames
        self.pcm = torch.from_numpy(pcm).cuda()
        self.ceps = torch.empty((self.F, self.nout), dtype=torch.float32, device="cuda")
        self.units = self.F

    def step(self):
        if self.gt:
            self.fe.run_batch_dev(self.off, self.pcm, self.ceps)
        else:
            self.fe.run_plan(self.plan, self.pcm, self.ceps)

    def epoch_reduce(self, world):
        pass

    def roofline(self):
        if self.gt:
            # sequential recursions (4 second-order sections per channel and sample, f32 in the reference's order): the parallel axes are
            # channels x segments only, so the kernel is bound by the latency of one dependent f32 chain per lane, not by bytes or flops
            ms, n = self.ctx.profile_get("gammatone")
            if n == 0:
                return None
            flops = self.F * 160.0 * 68 * (4 * 6 + 9)   # per sample and channel: 4 sections x 6 operations + temporal integration
            t = flops / (ms * 1e-3) / 1e12
            return dict(bound="valu", kernel="gammatone_filter_kernel (+ gammatone_post_kernel)", note="f32 vector operations against the f32 "
                        "vector peak; time-sequential IIR cascade, lane = (segment, channel)", achieved=round(t, 3), peak=FP32_TFLOPS,
                        unit="TFLOP/s", frac=round(t / FP32_TFLOPS, 5), traffic=None, avg_launch_ms=round(ms, 4), launches=n, flops_per_launch=flops)
        ms, n = self.ctx.profile_get("mfcc")
        if n == 0:
            return None
        lp, _ = self.ctx.profile_get("lpc_cepstrum")
        per_frame = 640.0 + 4.0 * self.nout   # 160 samples*4 B in + cepstra*4 B out per frame (800 B for MFCC-40)
        gbs = self.F * per_frame / ((ms + lp) * 1e-3) / 1e9
        return dict(bound="hbm", kernel="mfcc_kernel<256>" + (" + lpc_cepstrum_kernel" if self.plp else ""), achieved=round(gbs, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                    traffic=None if self.plp else measured_traffic("mfcc", "mfcc_kernel<256, 4>"), avg_launch_ms=round(ms + lp, 4),
                    launches=n, bytes_per_launch=self.F * per_frame)

    def stage_report(self):
        out = {}
        for k in ("mfcc", "lpc_cepstrum", "gammatone"):
            ms, n = self.ctx.profile_get(k)
            if n:
                out[k] = dict(avg_ms=round(ms, 4), launches=n)
        if not self.gt:
            try:
                out["end_to_end"] = self.end_to_end()
            except Exception as e:  # never take the line down
                out["end_to_end"] = dict(error=str(e)[:200])
        return out

    def end_to_end(self):
        """SURVEY 8(d) cfg 2 asks for the device-resident AND the host-link-inclusive rate (never `value`): the same batch with the
        samples (a) resident as s16 -- what the audio file holds, widened in the kernel like Flow/TypeConverter.hh:35-43 --, (b) copied
        from pinned host memory as f32 (640 B per frame over the link) and (c) as s16 (320 B per frame), copy and kernel on one stream"""
        import torch
        torch.cuda.synchronize()
        pcm16 = self.pcm.to(torch.int16)
        host32 = self.pcm.cpu().pin_memory()
        host16 = pcm16.cpu().pin_memory()
        dev32 = torch.empty_like(self.pcm)
        dev16 = torch.empty_like(pcm16)
        ref = torch.empty_like(self.ceps)
        self.fe.run_plan(self.plan, self.pcm, ref)

        def rate(fn, n=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return round(self.F * n / (time.perf_counter() - t0), 1)
        out = {"resident_s16_frames_per_s": rate(lambda: self.fe.run_plan(self.plan, pcm16, self.ceps))}
        out["s16_equals_f32_bitwise"] = bool(torch.equal(ref.view(torch.int32), self.ceps.view(torch.int32)))
        out["h2d_f32_frames_per_s"] = rate(lambda: (dev32.copy_(host32, non_blocking=True), self.fe.run_plan(self.plan, dev32, self.ceps)))
        out["h2d_s16_frames_per_s"] = rate(lambda: (dev16.copy_(host16, non_blocking=True), self.fe.run_plan(self.plan, dev16, self.ceps)))
        out["note"] = "pinned host PCM -> HBM -> cepstra in HBM; bytes over the link per frame: 640 (f32) / 320 (s16)"
        return out


class GmmOnly:
    def __init__(self, ctx, args, rank, tied):
        import torch

        import rasr_amd
        from tests import synth
        self.ctx, self.tied = ctx, tied
        if tied:
            model = synth.gmm_tied(10000, 4096, 40, seed=5, pooled=True)
        else:
            model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
        self.nk = int(model["mix_offsets"][-1])
        self.nd = len(model["dens_mean"])
        self.gmm_type = args.gmm_type
        self.sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type=args.gmm_type, tuning=args.gmm_tuning)
        self.T = int(getattr(args, "gmm_frames", 256))
        x = np.random.Generator(np.random.PCG64(4 + rank)).standard_normal((self.T, 40)).astype(np.float32)
        self.x = torch.from_numpy(x).cuda()
        self.scores = torch.empty((self.T, 10000), dtype=torch.float32, device="cuda")
        self.best = torch.empty((self.T, 10000), dtype=torch.int32, device="cuda")
        self.units = self.T

    def step(self):
        self.sc.score_dev(self.x, self.T, self.scores, None if self.gmm_type == "batch-diagonal-maximum-float" else self.best)

    def epoch_reduce(self, world):
        pass

    def roofline(self):
        # exact-order distance: sub, mul, mul, add per (frame, density, dim) = 4 f32 VALU ops; the f32 vector peak is
        # numerically the f32 MFMA peak (157.3 TFLOP/s counts an FMA as 2), so unfused ops can reach half of it.
        if self.tied:
            # combine stage: algorithmically one add + one compare/select per (frame, mixture, density).  The reference
            # does them in f64; gmm_tied_tile_kernel screens in f32 and runs the f64 rule on ~1 density per pair, so
            # the ops are priced against the f32 vector peak.  The distances (gmm_dist) are 0.5 MFLOP/frame, negligible
            ms, n = self.ctx.profile_get("gmm_combine")
            if n == 0:
                return None
            ops = 2.0 * self.nk * self.T
            surv, triples = self.sc.screen_counts(True)
            if triples:
                # pruned path (gmm_tied.hip): the time goes into reading rows of the 164 MB weight table at random -- 64 near rows per
                # frame for the bounds (whole rows of its bf16 image: every mixture) and one 256-byte row per surviving (density, frame, tile) triple --
                # plus the scores and density indices that leave.  `achieved` = those bytes / time of the four kernels.
                # `frac` follows SURVEY 8(d): the weight table once per batch (sum K_m x 4 B = 164 MB) + scores and density indices out
                # (8 B per frame and mixture) over the time of the scorer's kernels.  The rows the pruned kernel really touches -- 64 near
                # rows per frame for the bounds (whole rows of the bf16 image) and one 256-byte row per surviving (density, frame, tile)
                # triple, mostly L2 hits -- are reported next to it as l2_rows_GBps.
                launches = triples / float(4096 * self.T * 157) if self.T else 1.0
                rows = (self.T * 64.0 * 10048 * 2 + (surv / max(launches, 1.0)) * 256.0 + self.T * 10000 * 8.0 + self.T * 4096 * 12.0)
                by = self.nk * 4.0 + self.T * 10000 * 8.0 + self.T * 40 * 4.0
                ms_d, n_d = self.ctx.profile_get("gmm_dist")     # the distance kernel the `kernel` string names belongs to the time
                ms_combine, ms = ms, ms + (ms_d if n_d else 0.0)
                gbs = by / (ms * 1e-3) / 1e9
                return dict(bound="hbm", kernel="tied_pruned_kernel + tied_bound8_kernel + gmm_dist_list_kernel (+ tied_mask / tied_list; tied_near / tied_transpose off the list-order path)",
                            note="exact pruning: bounds from 64 near densities per frame, then the reference's f64 rule over the surviving "
                                 "(density, frame, 64-mixture tile) triples only; algorithmic bytes = weight table once per batch + results",
                            achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None,
                            avg_launch_ms=round(ms, 4), launches=n, bytes_per_launch=by,
                            time_is="gmm_dist (%.4f ms: distances and the near densities) + gmm_combine (%.4f ms: bound / list / mask / pruned kernels)" % (ms - ms_combine, ms_combine),
                            l2_rows_GBps=round(rows / (ms * 1e-3) / 1e9, 1),
                            surviving_fraction=round(surv / float(triples), 5),
                            dense_equivalent_tops=round(ops / (ms * 1e-3) / 1e12, 2),
                            algorithmic_speedup_vs_dense=round(ops / (ms * 1e-3) / 1e12 / FP32_TFLOPS, 3))
            ach = ops / (ms * 1e-3) / 1e12
            return dict(bound="valu", note="VALU tropical (min,+) contraction, 2 ops per (frame, mixture, density), priced against the f32 "
                                           "vector peak; not MFMA-able", kernel="gmm_tied_tile_kernel", achieved=round(ach, 3),
                        peak=FP32_TFLOPS, unit="TFLOP/s", frac=round(ach / FP32_TFLOPS, 4), traffic=None, avg_launch_ms=round(ms, 4),
                        launches=n, flops_per_launch=ops)
        elif self.gmm_type == "diagonal-maximum":
            return gmm_cart_roofline(self.ctx, self.sc, self.nk, 10000, 40, self.T)
        elif self.gmm_type == "SIMD-diagonal-maximum":
            # u8 means and features, integer distance: the products run on the i8 matrix pipes (2 x 160 000 x 64 ops per frame fit
            # 2.5 POP/s many times over), so the bound is the 8 bytes of (score, best density) per frame and mixture that leave
            ms, n = self.ctx.profile_get("gmm_simd")
            qs, _ = self.ctx.profile_get("gmm_simd_quantize")
            if n == 0:
                return None
            by = float(self.T) * (10000 * 8 + 40 * 4)
            ach = by / ((ms + qs) * 1e-3) / 1e9
            return dict(bound="hbm", kernel="simd_mfma_kernel (+ simd_quantize_kernel)", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, avg_launch_ms=round(ms + qs, 4), launches=n, bytes_per_launch=by,
                        note="algorithmic bytes = scores + best densities out (8 B per frame and mixture) + features in")
        else:
            ms, n = self.ctx.profile_get("gmm")
            ops = 3.0 * self.nk * 40 * self.T   # batch-float: pre-scaled means, sub/mul/add
            name = "gmm_batch_float_kernel<40>"
        if n == 0:
            return None
        ach = ops / (ms * 1e-3) / 1e12
        return dict(bound="valu", note="f32 VALU kernel priced against the f32 vector peak", kernel=name,
                    achieved=round(ach, 3), peak=FP32_TFLOPS, unit="TFLOP/s", frac=round(ach / FP32_TFLOPS, 4), traffic=None,
                    avg_launch_ms=round(ms, 4), launches=n, flops_per_launch=ops)

    def stage_report(self):
        out = {}
        for k in ("gmm_screen_pack", "gmm_screen", "gmm", "gmm_dist", "gmm_combine", "gmm_simd_quantize", "gmm_simd", "gmm_simd_dist"):
            ms, n = self.ctx.profile_get(k)
            if n:
                out[k] = dict(avg_ms=round(ms, 4), launches=n)
        return out


class NnOnly:
    def __init__(self, ctx, args, rank):
        import torch

        import rasr_amd
        from tests import synth
        self.ctx = ctx
        dims = [440] + [2048] * 6 + [10000]
        Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
        self.nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision=args.precision, tuning=args.nn_tuning)
        self.nn_precision = self.nn.effective_precision()[0]   # (f16mx on heavy-tailed weights computes in split bf16: amx_ffnn_precision)
        self.requested_precision = args.precision
        self.T = 1024
        x = np.random.Generator(np.random.PCG64(6 + rank)).standard_normal((self.T, 440)).astype(np.float32)
        self.x = torch.from_numpy(x).cuda()
        self.scores = torch.empty((self.T, 10000), dtype=torch.float32, device="cuda")
        self.units = self.T
        self.flops = 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) * self.T

    def step(self):
        self.nn.score_dev(self.x, 440, self.T, self.scores)

    def epoch_reduce(self, world):
        pass

    def roofline(self):
        ms, n = self.ctx.profile_get("ffnn_gemm_max")
        if n == 0:
            return None
        out = nn_gemm_roofline(self.nn_precision, ms, n, self.T, False)
        # the whole network (7 GEMMs + packing), not only its largest layer: matrix-pipe time of all layers / time of the pass
        ms_all, n_all = self.ctx.profile_get("ffnn_gemm")
        ms_pk, n_pk = self.ctx.profile_get("ffnn_pack")
        if n_all:
            per_pass = ms_all * n_all / n + (ms_pk * n_pk / n if n_pk else 0.0)
            out["whole_network"] = self.whole_network(per_pass, "sum of the pass's kernels, plain launches (HIP events)")
        return out

    def whole_network(self, ms_per_pass, what):
        mult = {"bf16x3": 3.0, "f16mx": 1.5}.get(self.nn_precision, 1.0)
        peak = FP32_TFLOPS if self.nn_precision == "fp32" else MFMA_BF16_TFLOPS
        ach = mult * self.flops / (ms_per_pass * 1e-3) / 1e12
        return dict(achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4), ms_per_pass=round(ms_per_pass, 4), time_is=what,
                    algorithmic_tflops=round(self.flops / (ms_per_pass * 1e-3) / 1e12, 2))

    def stage_report(self):
        out = {}
        for k in ("ffnn_pack", "ffnn_gemm", "ffnn_gemm_max", "ffnn_split"):
            ms, n = self.ctx.profile_get(k)
            if n:
                out[k] = dict(avg_ms=round(ms, 4), launches=n)
        return out


class NullJob:
    """Host-only stand-in for a rank's work (--workload null, --backend gloo): no kernels, no GPU.  It exists so that the launcher,
    the rendezvous, the reference's partition rule, the one-collective epoch reduce and the JSON line of the N-rank run can be
    exercised where there is no GPU (tests/test_distributed.py).  A "frame" is a counter increment; the number it reports says
    nothing about the product."""

    N_STATES = 16

    def __init__(self, args, rank, world):
        from rasr_amd.partition import EpochReduceBuffer, select_partition
        self.n_total = args.utterances * world                       # the corpus of this step: weak scaling
        self.mine = select_partition(self.n_total, world, rank)      # segment i -> rank i % world
        self.frames = 100 + 7 * (np.asarray(self.mine) % 5)          # frames of my utterances
        self.units = int(self.frames.sum())
        self.red = EpochReduceBuffer([("score_sum", 1, "f64"), ("counts", self.N_STATES, "count"), ("frames", 1, "count")], device="cpu")
        self.steps_done = 0
        # --ingest streamed: the corpus walk of StreamedIngest without the copies -- which utterances this rank would stream
        self.walker, self.visited, self.world, self.rank = None, [], world, rank
        if ingest_mode(args, world) == "streamed":
            from rasr_amd.partition import CorpusWalker
            self.n_corpus = max(int(args.corpus_hours * 3600.0 / args.utt_seconds), args.utterances * world)
            self.walker = CorpusWalker(self.n_corpus, world, rank, args.utterances)

    def step(self):
        import torch
        if self.walker is not None:
            self.visited.append(self.walker.next_batch())
        self.red.view("counts").add_(torch.from_numpy(np.bincount(np.asarray(self.mine) % self.N_STATES, weights=self.frames,
                                                                  minlength=self.N_STATES).astype(np.int64)))
        self.red.view("frames").add_(int(self.units))
        self.red.view("score_sum").add_(float(self.units) * 0.5)
        self.steps_done += 1

    def epoch_reduce(self, world):
        self.red.all_reduce()   # torch.distributed (gloo): the CPU stand-in of amx_comm_all_reduce_f64_dev

    def roofline(self):
        return None

    def stage_report(self):
        # what every rank must agree on after the reduce: all utterances of all ranks, every step
        all_frames = 100 + 7 * (np.arange(self.n_total) % 5)
        want = int(all_frames.sum()) * self.steps_done
        got = int(self.red.view("frames")[0])
        out = {"reduced_frames": got, "expected_frames": want, "reduce_ok": bool(got == want and int(self.red.view("counts").sum()) == want)}
        if self.walker is not None:
            out["corpus_walk"] = self.walk_report()
        return out

    def gather_walks(self):
        """every rank's visited list on every rank (control plane, gloo); called by all ranks"""
        import torch.distributed as dist
        mine = np.concatenate(self.visited).tolist() if self.visited else []
        self.walks = [mine]
        if dist.is_available() and dist.is_initialized():
            self.walks = [None] * self.world
            dist.all_gather_object(self.walks, mine)

    def walk_report(self):
        allv = np.concatenate([np.asarray(w, dtype=np.int64) for w in self.walks]) if self.walks else np.zeros(0, np.int64)
        per_rank_ok = all(all(int(g) % self.world == r for g in w) for r, w in enumerate(self.walks))
        return {"corpus_utterances": self.n_corpus, "visited": int(len(allv)), "distinct": int(len(np.unique(allv))),
                "ranks_disjoint": bool(len(np.unique(allv)) == len(allv) or self.walker.epoch > 0),
                "every_rank_in_its_partition": bool(per_rank_ok), "first_of_each_rank": [w[:3] for w in self.walks]}


# ----------------------------------------------------------------------------------------------- CPU baseline

def _cpu_mfcc_worker(job):
    """runs in a spawned process: returns (frames, seconds of compute) for its utterances, excluding start-up"""
    from oracle import OracleMfcc
    from oracle.binding import set_default_contract, use_native_oracle
    seeds, reps, native, contract = job
    from tests import synth
    set_default_contract(contract)
    if native:
        use_native_oracle()
    pcms = [synth.waveform(160000, seed=s) for s in seeds]
    m = OracleMfcc(n_ceps=40, filter_width=138.0)
    m.run(pcms[0][:16000])  # warm
    n = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        for p in pcms:
            n += m.run(p).shape[0]
    return n, time.perf_counter() - t0


def _cpu_gmm_worker(job):
    """runs in a spawned process: (frames, seconds) of one CPU GMM scorer of the oracle on the config-3 model.
    kind "frame": Mm::GaussDiagonalMaximumFeatureScorer loop (one frame at a time, a13); "batch": Mm::BatchFloatFeatureScorer
    (a18, pooled covariance, pre-scaled means -- the reference's fastest CPU scorer), model preparation excluded by differencing
    a T-frame and a 1-frame call; "tied": the per-frame loop on the tied 4096 x 10000 model."""
    kind, T, native, contract = job
    from oracle import OracleGmm
    from oracle.binding import set_default_contract
    from tests import synth
    set_default_contract(contract)
    if native:
        from oracle.binding import use_native_oracle
        use_native_oracle()
    if kind == "tied":
        model = synth.gmm_tied(10000, 4096, 40, seed=5, pooled=True)
    else:
        model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
    x = np.random.Generator(np.random.PCG64(4)).standard_normal((T, 40)).astype(np.float32)
    g = OracleGmm(model)
    if kind == "batch":
        g.score_batch_float(x[:1])
        t0 = time.perf_counter()
        g.score_batch_float(x[:1])
        t1 = time.perf_counter()
        g.score_batch_float(x)
        t2 = time.perf_counter()
        return T - 1, max((t2 - t1) - (t1 - t0), 1e-9)
    g.score(x[:1], mode=0, want_best=False)  # warm
    t0 = time.perf_counter()
    g.score(x, mode=0, want_best=True)
    return T, time.perf_counter() - t0


def _cpu_nn(threads, seconds):
    """numpy float32 matmul = OpenBLAS sgemm with separate bias / ReLU passes like Nn::LinearLayer + ActivationLayer
    (Nn/LinearLayer.cc:298-324, Math/Blas.hh:402-420), limited to `threads` BLAS threads"""
    from tests import synth
    dims = [440] + [2048] * 6 + [10000]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
    T = 8192 if threads > 1 else 512
    x = np.random.Generator(np.random.PCG64(6)).standard_normal((T, 440)).astype(np.float32)
    WT = [np.ascontiguousarray(w.T) for w in Ws]
    bl = bs[-1] - np.float32(1.0) * logp

    def fwd():
        a = x
        for l in range(len(Ws)):
            z = a @ WT[l]                 # sgemm
            z += (bl if l == len(Ws) - 1 else bs[l])   # addToAllColumns
            if l < len(Ws) - 1:
                np.maximum(z, 0, out=z)   # ensureMinimalValue(0)
            a = z
        return -a
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=threads, user_api="blas")
    except Exception:
        limit = None
    try:
        fwd()
        reps = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            fwd()
            reps += 1
        dt = time.perf_counter() - t0
    finally:
        if limit is not None:
            limit.restore_original_limits()
    return T * reps, dt


def cpu_baseline(workload, contract="fma"):
    """The oracle (CPU restatement of the reference path, kind "port") timed on a bounded sample of the same workload on the
    host's cores.  The oracle is rebuilt for this host (-O3 -march=native) into a temporary directory, in the run's contract mode: with
    contract=fma its accumulates are vfmadd instructions -- the instruction mix of the reference's DEFAULT build (-march=native, GCC's
    -ffp-contract=fast) -- with contract=off they are separate products and sums like a -DMARCH=x86-64 build; nothing else is
    contracted in either (the results are those of the checker library of the same mode).  Every variant SURVEY 8(d) lists is timed and named in
    `sample`; `value` combines the FASTEST variant of each stage (the honest comparison):
      MFCC   frame by frame, one process per hardware thread
      GMM    (i) Mm::GaussDiagonalMaximumFeatureScorer per frame, all threads; (ii) Mm::BatchFloatFeatureScorer (a18), all threads
      NN     OpenBLAS sgemm + separate bias / ReLU passes at 1 thread and at all threads"""
    import multiprocessing as mp

    cores = os.cpu_count() or 1
    res, notes, variants = {}, [], {}
    native = True
    try:
        from oracle.binding import build_native_oracle
        build_native_oracle(contract)
    except Exception as e:  # no compiler on the box: the -O2 library that travelled with the repository
        native = False
        notes.append("native oracle build failed (%s): -O2 build used" % str(e)[:60])
    pool = lambda: mp.get_context("spawn").Pool(cores)
    if workload in ("pipeline", "nn-pipeline", "mfcc"):
        with pool() as p:
            out = p.map(_cpu_mfcc_worker, [([1000 + i], 4, native, contract) for i in range(cores)])   # 4 x 10 s of audio per process
        res["mfcc"] = (sum(o[0] for o in out), max(o[1] for o in out))
        notes.append("MFCC: %d oracle processes x 40 s audio, %.2f s" % (cores, res["mfcc"][1]))
    if workload in ("pipeline", "gmm", "gmm-train"):
        with pool() as p:
            a = p.map(_cpu_gmm_worker, [("frame", 8, native, contract)] * cores)
        with pool() as p:
            b = p.map(_cpu_gmm_worker, [("batch", 129, native, contract)] * cores)
        va = (sum(o[0] for o in a), max(o[1] for o in a))
        vb = (sum(o[0] for o in b), max(o[1] for o in b))
        variants["gmm diagonal-maximum per frame, %d threads" % cores] = va[0] / va[1]
        variants["gmm batch-diagonal-maximum-float (a18), %d threads" % cores] = vb[0] / vb[1]
        res["gmm"] = va if va[0] / va[1] >= vb[0] / vb[1] else vb
    if workload == "gmm-tied":
        n = min(cores, 32)   # 164 MB of weights per process
        with mp.get_context("spawn").Pool(n) as p:
            a = p.map(_cpu_gmm_worker, [("tied", 1, native, contract)] * n)
        res["gmm"] = (sum(o[0] for o in a), max(o[1] for o in a))
        variants["gmm tied 4096 x 10000 diagonal-maximum per frame, %d threads" % n] = res["gmm"][0] / res["gmm"][1]
        cores = n
    if workload in ("pipeline", "nn-pipeline", "nn"):
        one = _cpu_nn(1, 3.0)
        allc = _cpu_nn(cores, 6.0)
        variants["nn sgemm 1 thread"] = one[0] / one[1]
        variants["nn sgemm %d threads" % cores] = allc[0] / allc[1]
        res["nn"] = allc if allc[0] / allc[1] >= one[0] / one[1] else one
    # frames/s of the whole CPU job = 1 / sum(stage seconds per frame), fastest variant of every stage
    spf = sum(dt / fr for fr, dt in res.values())
    detail = ", ".join("%s %d frames in %.2fs" % (k, fr, dt) for k, (fr, dt) in res.items())
    vtxt = "; ".join("%s: %.1f frames/s" % (k, v) for k, v in variants.items())
    return dict(value=round(1.0 / spf, 2), unit="frames/s", cores=cores, kind="port",
                sample=detail + " (" + "; ".join(notes + [vtxt]) + "; oracle built %s, contract=%s)" % ("-O3 -march=native" if native else "-O2", contract))


def parity_check(ctx, job, args, n_nn=256, n_gmm=24):
    """Part of the cpu_baseline leg: the oracle as the CHECKER of this run's scorers on a sample of the frames the timed steps
    scored (the first frames of the last batch: real cepstra / context windows of the job, not a separate input).
      nn:  every score of n_nn frames against the f64-accumulating oracle (Nn/LinearLayer.cc:298-324 arithmetic with exact sums) --
           worst error over the 1e-4 |ref| + 1e-4 bar, worst PURE relative error over |ref| > 1e-2, arg-min mismatches over all
           sampled frames against the oracle and against the library's exact-f32 MFMA path, frames a 1e-5 gap rule would exclude
           (tests/parity.py; the contract is what a decoder reads, Nn/BatchFeatureScorer.cc:92-171)
      gmm: scores and best-density indices of n_gmm frames x 10 000 mixtures, bit for bit (Mm/GaussDiagonalMaximumFeatureScorer.cc:116-180)"""
    import torch

    import rasr_amd
    from oracle import OracleGmm, oracle_ffnn_score
    from tests import synth
    from tests.parity import nn_parity_report
    out = {}
    torch.cuda.synchronize()
    if hasattr(job, "nn"):
        dims = [440] + [2048] * 6 + [10000]
        Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
        x = job.ctxwin[:n_nn].contiguous()
        got = torch.empty((n_nn, 10000), dtype=torch.float32, device="cuda")
        job.nn.score_dev(x, 440, n_nn, got)
        f32 = torch.empty_like(got)
        rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="fp32").score_dev(x, 440, n_nn, f32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        want = oracle_ffnn_score(Ws, bs, acts, x.cpu().numpy(), log_prior=logp, prior_scale=1.0, acc64=True)
        rep = nn_parity_report(got.cpu().numpy(), want, other=f32.cpu().numpy(), gap=1e-5)
        rep.update(precision=args.precision, reference="oracle_ffnn_score(acc64=True) on the job's own context windows",
                   oracle_seconds=round(time.perf_counter() - t0, 2))
        out["nn"] = rep
    g = getattr(job, "gmm", None) or getattr(job, "sc", None)
    if g is not None:
        model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
        x = job.ceps[:n_gmm].contiguous()
        sc = torch.empty((n_gmm, 10000), dtype=torch.float32, device="cuda")
        bd = torch.empty((n_gmm, 10000), dtype=torch.int32, device="cuda")
        g.score_dev(x, n_gmm, sc, bd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ws, wb = OracleGmm(model, contract=args.contract).score(x.cpu().numpy(), mode=0, want_best=True)
        gs, gb = sc.cpu().numpy(), bd.cpu().numpy()
        out["gmm"] = dict(frames=n_gmm, scores=int(gs.size), score_bit_mismatches=int((gs.view(np.uint32) != ws.view(np.uint32)).sum()),
                          best_density_mismatches=int((gb.astype(np.int64) != wb.astype(np.int64)).sum()),
                          reference="OracleGmm.score (orc_score.c, diagonal-maximum, contract=%s) on the job's own cepstra" % args.contract,
                          oracle_seconds=round(time.perf_counter() - t0, 2))
    return out


def make_job(ctx, args, rank, world=1):
    args._world = world
    if args.workload == "null":
        return NullJob(args, rank, world)
    if args.workload in ("pipeline", "nn-pipeline"):
        job = (Pipeline if args.workload == "pipeline" else NnPipeline)(ctx, args, rank)
        # what the handle COMPUTES in: f16mx requested on heavy-tailed weights runs split bf16 (amx_ffnn_precision; mx_fallback=auto) --
        # the roofline's kernel name and executed units follow the effective precision, the line names both
        job.nn_precision = job.nn.effective_precision()[0] if hasattr(job.nn, "effective_precision") else args.precision
        job.requested_precision = args.precision
    elif args.workload == "mfcc":
        job = MfccOnly(ctx, args, rank)
    elif args.workload == "gmm-train":
        job = GmmTrain(ctx, args, rank)
    elif args.workload == "gmm-trained":
        job = GmmTrained(ctx, args, rank)
    elif args.workload in ("gmm", "gmm-tied"):
        job = GmmOnly(ctx, args, rank, tied=args.workload == "gmm-tied")
    else:
        job = NnOnly(ctx, args, rank)
    return job


def is_graph_mode(args):
    # tuning graph=1 (opt-in since round 6: plain launches measured 3-5 % faster in this loop): config 4 (batch 1024) and the config-3
    # scorers (batch 256) replay their pass as a HIP graph, which the per-launch events of the library's profiler would switch off
    return (args.workload == "nn" and "graph=1" in (args.nn_tuning or "")) or \
           (args.workload in ("gmm", "gmm-tied") and args.gmm_type == "diagonal-maximum" and args.gmm_frames <= 4096
            and "graph=1" in (args.gmm_tuning or ""))


def reset_survivor_counters(job):
    for name in ("gmm", "sc"):  # survivor counter of the fused GMM scorer (one atomic per wavefront and launch)
        g = getattr(job, name, None)
        if g is not None and hasattr(g, "screen_counts"):
            g.screen_counts(True)


SHADER_CLOCK_GHZ = None   # set by measure(): median shader clock of the eight XCDs over the second (per-launch timed) pass
SHADER_CLOCK_XCD = None
HWMON = None              # set by measure(): the driver's own sclk / socket power (rocm-smi) over a stretch of the same steps


class SmiSampler:
    """sclk and socket power as `rocm-smi --showpower --showclocks` reports them (the driver's gpu_metrics table), polled from a thread
    while a stretch of work runs -- one call takes a few hundred milliseconds, so the stretch has to last seconds.  (The hwmon files
    freq1_input / power1_input of this driver lag by seconds: 0.11 GHz / 241 W in the middle of a pass that rocm-smi puts at 1.95 GHz /
    1350 W.)  Host-side only: nothing is enqueued on the device.  Best effort -- no rocm-smi, no entry in the line."""

    def __init__(self, local_rank=0):
        import shutil
        self.exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self.gpu, self.f, self.p, self.stop, self.th = "GPU[%d]" % local_rank, [], [], False, None

    def _run(self):
        import re
        import subprocess
        while not self.stop:
            try:
                out = subprocess.run([self.exe, "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                f = p = None
                for ln in out.splitlines():
                    if not ln.startswith(self.gpu):
                        continue
                    m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", ln)
                    if m:
                        f = int(m.group(1))
                    m = re.search(r"Power \(W\):\s*([\d.]+)", ln)
                    if m:
                        p = float(m.group(1))
                if f is not None and p is not None:
                    self.f.append(f)
                    self.p.append(p)
            except Exception:
                time.sleep(0.2)

    def __enter__(self):
        if self.exe:
            import threading
            self.th = threading.Thread(target=self._run, daemon=True)
            self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.th:
            self.th.join()

    def report(self, what):
        f, p = self.f[1:], self.p[1:]   # the first sample may predate the load
        if not f:
            return None
        return dict(sclk_GHz=round(float(np.mean(f)) / 1e3, 3), sclk_GHz_min_max=[round(min(f) / 1e3, 3), round(max(f) / 1e3, 3)],
                    socket_power_W=round(float(np.mean(p)), 1), power_cap_W=1400.0, samples=len(f),
                    source="rocm-smi --showpower --showclocks, back to back during " + what)


def smi_stretch(job):
    """~2.5 s more of the same steps with rocm-smi polled beside them: the clock and the socket power the workload settles at (the
    timed region and the per-launch pass are too short for a tool that takes 0.1-0.3 s per reading).  Called AFTER the roofline and the
    stage report have been read: the device-side survivor counters keep counting here."""
    import torch
    global HWMON
    if os.environ.get("AMX_BENCH_NO_SMI", "0") not in ("", "0"):
        return
    n = 0
    t1 = time.perf_counter()
    with SmiSampler(int(os.environ.get("LOCAL_RANK", "0"))) as smi:
        if smi.exe:
            while time.perf_counter() - t1 < 2.5:
                job.step()
                n += 1
                if n % 8 == 0:
                    torch.cuda.synchronize()   # keep the host at most a few steps ahead: the stretch ends when the clock says so
            torch.cuda.synchronize()
    HWMON = smi.report("%d more steps of the same workload (%.1f s) behind the timed region" % (n, time.perf_counter() - t1))


def measure(ctx, job, args, world):
    """W untimed steps, then exactly K steps + the epoch reduce between barrier + synchronize; returns seconds"""
    gpu = ctx is not None
    for _ in range(args.warmup):
        job.step()
    barrier(world, gpu)
    if gpu:
        ctx.profile(False)   # the timed region runs WITHOUT the per-launch events (two hipEventRecord per launch); see below
        ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    job.epoch_reduce(world)
    barrier(world, gpu)
    dt = time.perf_counter() - t0
    if gpu and getattr(job, "counts", None) is not None and hasattr(job, "red") and not hasattr(job, "reduced_frames"):   # the headline's pass only (the streamed comparison times the same job again)
        # integrity of the one collective, read BEHIND the timed region: after the reduce every rank holds the frames of ALL ranks
        # (each frame adds one to its best state's counter in the warm-up and the timed steps alike)
        job.reduced_frames = int(job.counts.sum().item())
        job.expected_frames = int(job.units) * (args.warmup + args.steps) * world
    if gpu:
        # kernel timings (roofline, stages) come from a SECOND pass over the same steps with the per-launch HIP events on (for a
        # graph-replayed workload: with plain launches).  The survivor counter is reset WITH the profiler, so that it covers exactly
        # the launches the events cover.
        import torch
        reset_survivor_counters(job)
        ctx.profile(True)
        ctx.profile_reset()
        # shader clock of the same pass: (s_memtime, s_memrealtime) sampled by a one-wave kernel in front of and behind it, in stream
        # order.  The NN workloads sit at the package's 1400 W power cap and the firmware lowers sclk until they fit
        # (profiles/r05/power_probe.log): `peak` in the roofline object is priced at the 2.4 GHz ceiling, this says what the run got.
        clk = torch.zeros((2, 8, 2), dtype=torch.int64, device="cuda")
        ctx.device_clocks_xcd(clk[0])
        for _ in range(min(args.steps, 20)):
            job.step()
        ctx.device_clocks_xcd(clk[1])
        torch.cuda.synchronize()
        ck = clk.cpu().numpy().astype(np.float64)
        global SHADER_CLOCK_GHZ, SHADER_CLOCK_XCD
        ok = (ck[0, :, 1] > 0) & (ck[1, :, 1] > ck[0, :, 1])   # XCDs both samples reached
        per = (ck[1, :, 0] - ck[0, :, 0]) / np.maximum(ck[1, :, 1] - ck[0, :, 1], 1.0) * 0.1
        if ok.any():
            SHADER_CLOCK_GHZ = round(float(np.median(per[ok])), 3)   # the median: single XCDs read 1.5 or 2.9 GHz over 160 ms (s_memtime counts per CU, unaligned)
            SHADER_CLOCK_XCD = [round(float(v), 3) if o else None for v, o in zip(per, ok)]
    if gpu:
        ctx.profile(False)
    return dt


WORKLOAD_NAMES = {
    "pipeline": lambda a: "cfg5-shard: MFCC-40 -> {GMM 10000x16 diagonal-maximum (scores f32 of all states; best densities: " + {"u32": "u32 matrix, all states", "u8": "byte matrix, all states", "aligned": "of each frame's aligned (= best) state, amx_gmm_best_density_dev"}[a.best_density] + ") -> Viterbi accumulators | ctx11 -> FFNN 440-6x2048-10000 "
                          "(%s MFMA) -> best-state counts}, every frame scored by both models; %d utterances x %.0f s per step and rank"
                          % (a.precision, a.utterances, a.utt_seconds),
    "nn-pipeline": lambda a: "cfg5-shard, NN leg only: MFCC-40 -> ctx11 -> FFNN 440-6x2048-10000 (%s MFMA) -> best-state counts; "
                             "%d utterances x %.0f s per step and rank" % (a.precision, a.utterances, a.utt_seconds),
    "mfcc": lambda a: "cfg2: batched MFCC-40 on 1000 utterances (5-15 s)" if getattr(a, "front_end", "mfcc") == "mfcc"
                      else "cfg2 audio through %s.flow: 1000 utterances (5-15 s)" % a.front_end,
    "gmm": lambda a: "cfg3-cart: 10000 states x 16 densities, d=40, pooled covariance, batch %d, %s" % (a.gmm_frames, a.gmm_type),
    "gmm-tied": lambda a: "cfg3-tied: 4096 shared densities x 10000 states, d=40, batch %d, diagonal-maximum" % a.gmm_frames,
    "nn": lambda a: "cfg4: FFNN 440-6x2048-10000 (%s MFMA), batch 1024" % a.precision,
    "gmm-trained": lambda a: "cfg5 GMM leg on a split-trained model: 10000 states grown 1 -> 16 densities by estimate + split + Viterbi "
                             "re-estimation on clustered synthetic features (pooled covariance), 63936 frames of those features per step",
    "gmm-train": lambda a: "cfg5 GMM leg: MFCC-40 -> 10000x16 GMM (diagonal-maximum) -> Viterbi accumulators (f64) ; "
                           "%d utterances x %.0f s per step and rank" % (a.utterances, a.utt_seconds)}


def streamed_comparison(ctx, job, args, resident_value):
    """Single-GPU line: the same job once more with its audio streamed (StreamedIngest) and once with the batch resident as s16 --
    the streamed run's kernel -- so that the cost of the ingest is on record next to the resident `value`.  Called after the
    roofline / stage numbers of the headline have been read (the profiler is reset here)."""
    a = copy_args(args, ingest="streamed")
    ing = StreamedIngest(a, job.rank, 1, a.steps + a.warmup)
    job.ingest = ing
    dt = measure(ctx, job, a, 1)
    streamed = job.units * a.steps / dt
    visited = np.concatenate(ing.visited)
    job.ingest = None
    pcm32 = job.pcm
    job.pcm = pcm32.to(job.torch_mod().int16) if hasattr(job, "torch_mod") else pcm32
    dt16 = measure(ctx, job, a, 1)
    job.pcm = pcm32
    res16 = job.units * a.steps / dt16
    out = ing.report()
    del out["mode"]
    out.update(frames_per_s=round(streamed, 1), ms_per_step=round(1e3 * dt / a.steps, 4), resident_s16_frames_per_s=round(res16, 1),
               streamed_over_resident=round(streamed / resident_value, 4), streamed_over_resident_s16=round(streamed / res16, 4),
               link_GBps=round(ing.bytes_per_batch * a.steps / dt / 1e9, 3),
               utterances_visited=int(len(visited)), distinct_utterances_visited=int(len(np.unique(visited))))
    return out


def copy_args(args, **over):
    import copy
    a = copy.copy(args)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def secondary_configs(ctx, args, rank):
    """The other BASELINE configs and the plain-bf16 variant of the headline, measured in the same process (rank 0, one GPU):
    short runs of the single-stage workloads so that the driver's record carries them next to the headline."""
    import copy
    import gc

    import torch
    out = {}
    # (the sub-millisecond configs -- 3 and 4 -- get hundreds of warm-up and timed steps: each job is built on the host for seconds while the
    # chip idles, and 5 warm-up steps of 0.2 ms ended inside the clock ramp: config 4 read 0.262 ms here and 0.238 ms as `--workload nn
    # --steps 200 --warmup 20` on the same box, profiles/r06)
    plan = [("cfg5-shard with the NN in plain bf16 (BASELINE config 4's literal dtype; scores 2e-3-grade, NOT within north_star's 1e-4)",
             dict(workload="pipeline", precision="bf16", steps=20, warmup=2)),
            ("cfg2 mfcc", dict(workload="mfcc", steps=20, warmup=2)),
            ("cfg3 gmm-tied (4096 shared densities x 10000 states, batch 256)", dict(workload="gmm-tied", steps=200, warmup=100)),
            ("cfg3 gmm-cart (10000 x 16 densities, batch 256)", dict(workload="gmm", steps=400, warmup=200)),
            ("cfg5-shard GMM leg, split-trained model (10000 states grown 1 -> 16 densities by the repository's training loop)",
             dict(workload="gmm-trained", steps=8, warmup=2)),
            ("cfg5-shard GMM leg, random-init model (the headline's GMM leg alone)", dict(workload="gmm-train", steps=8, warmup=2)),
            ("cfg5-shard with the NN in split bf16 (round 3's default, the same 1e-4 bar at 3 MFMA products per product)",
             dict(workload="pipeline", precision="bf16x3", steps=20, warmup=2)),
            ("cfg4 nn f16mx (batch 1024)", dict(workload="nn", precision="f16mx", steps=400, warmup=200)),
            ("cfg4 nn bf16x3 (batch 1024)", dict(workload="nn", precision="bf16x3", steps=400, warmup=200)),
            ("cfg4 nn bf16 (batch 1024)", dict(workload="nn", precision="bf16", steps=400, warmup=200))]
    try:   # config 5 at N = 1 as one epoch (7-8 s): first, while the chip is cool -- its last second is the sustained figure
        out["cfg5 full epoch on one GPU (100 h = 36 000 utterances streamed, every frame through both models, ONE reduce at the end)"] = full_epoch(ctx, args, rank)
    except Exception as e:
        out["cfg5 full epoch on one GPU"] = dict(error=str(e)[:300])
    gc.collect()
    torch.cuda.empty_cache()
    for name, over in plan:
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        try:
            job = make_job(ctx, a, rank)
            dt = measure(ctx, job, a, 1)
            r = job.roofline() or {}
            out[name] = dict(value=round(job.units * a.steps / dt, 1), unit="frames/s", ms_per_step=round(1e3 * dt / a.steps, 4),
                             kernel=r.get("kernel"), roofline_frac=r.get("frac"), roofline_bound=r.get("bound"),
                             workload=WORKLOAD_NAMES[a.workload](a))
            if a.workload == "nn":
                # config 4: `roofline_frac` is the WHOLE network against the wall time of a pass; the largest
                # layer's own figure stays next to it
                wn = job.whole_network(1e3 * dt / a.steps, "wall time of one pass")
                out[name].update(roofline_frac=wn["frac"], output_layer_frac=r.get("frac"), whole_network=wn,
                                 kernel="all 7 GEMMs of the pass; largest: " + str(r.get("kernel")))
            if r.get("time_is"):
                out[name]["time_is"] = r["time_is"]
            if a.workload in ("gmm-trained", "gmm-train"):
                out[name]["survivors_per_mixture"] = r.get("survivors_per_mixture")
                out[name]["gmm_kernel_ms"] = r.get("avg_launch_ms")
                if a.workload == "gmm-trained":
                    st = job.stage_report()
                    out[name].update(model=st.get("model"), dense_kernel_ms=st.get("dense_kernel_ms"),
                                     screened_equals_dense_bitwise=st.get("screened_equals_dense_bitwise"), just_split_twins=st.get("just_split_twins"))
        except Exception as e:  # a secondary line must never take the headline down
            out[name] = dict(error=str(e)[:200])
        job = None
        gc.collect()
        torch.cuda.empty_cache()
    return out


def full_epoch(ctx, args, rank):
    """BASELINE config 5 run as what it says, on ONE GPU: a whole epoch over the 100 h corpus (36 000 utterances of 10 s: 563 batches of
    64), every batch streamed from pinned host memory, every frame scored by both models, the accumulators reduced ONCE at the end.  This
    is the job a rank of an 8-GPU run executes on its partition.  Reports what a 20-step figure cannot: frames/s over the first and over
    the last second of the epoch, the shader clock the chip sustains at both ends (s_memtime ticks per s_memrealtime tick, sampled in
    stream order), and the real-time factor as the reference defines it (Speech/CorpusProcessor.cc:49-58: wall time / audio time)."""
    import copy

    import torch
    a = copy.copy(args)
    a.workload, a.ingest, a.steps, a.warmup = "pipeline", "streamed", 22, 0   # host window: 24 batches of the partition (it wraps)
    job = make_job(ctx, a, rank, 1)
    ing = job.ingest
    for _ in range(3):   # untimed: workspaces, graphs, the first copies
        job.step()
    torch.cuda.synchronize()
    ing.walker.pos, ing.walker.epoch, ing.visited = 0, 0, []
    job.red.zero()
    n_steps = ing.walker.batches_per_epoch()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
    clk = torch.zeros((4, 8, 2), dtype=torch.int64, device="cuda")
    marks = {8: 0, min(n_steps - 1, 70): 1, max(9, n_steps - 70): 2, n_steps - 1: 3}   # ~1 s apart at ~14 ms per step
    ctx.profile(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    with SmiSampler(int(os.environ.get("LOCAL_RANK", "0"))) as hw:
        for s_ in range(n_steps):
            job.step()
            ev[s_ + 1].record()
            if s_ in marks:
                ctx.device_clocks_xcd(clk[marks[s_]])
        job.epoch_reduce(1)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    t_ms = np.array([ev[0].elapsed_time(e) for e in ev[1:]])          # completion time of every step on the device
    real = [len(v) for v in ing.visited[:n_steps]]                    # utterances of the corpus in each batch (the last one is short)
    frames = np.array([u * (job.F // a.utterances) for u in real], np.float64)
    total_frames, audio_s = float(frames.sum()), float(sum(real)) * a.utt_seconds
    first = t_ms <= 1000.0
    last = t_ms > t_ms[-1] - 1000.0
    ck = clk.cpu().numpy().astype(np.float64)
    def ghz(i, j):   # ticks per 10 ns -> GHz, median over the XCDs both samples reached
        ok = (ck[i, :, 1] > 0) & (ck[j, :, 1] > ck[i, :, 1])
        per = (ck[j, :, 0] - ck[i, :, 0]) / np.maximum(ck[j, :, 1] - ck[i, :, 1], 1.0) * 0.1
        return round(float(np.median(per[ok])), 3) if ok.any() else None

    def ghz_xcd(i, j):   # the same per XCD (over seconds the offset between the CUs that took the samples is below 0.1 %)
        ok = (ck[i, :, 1] > 0) & (ck[j, :, 1] > ck[i, :, 1])
        per = (ck[j, :, 0] - ck[i, :, 0]) / np.maximum(ck[j, :, 1] - ck[i, :, 1], 1.0) * 0.1
        return [round(float(v), 3) if o else None for v, o in zip(per, ok)]
    return dict(value=round(total_frames / wall, 1), unit="frames/s", wall_s=round(wall, 3), steps=n_steps, utterances=int(sum(real)),
                frames=int(total_frames), audio_hours=round(audio_s / 3600.0, 2), rtf=round(wall / audio_s, 8),
                rtf_definition="wall time / audio time (Speech/CorpusProcessor.cc:49-58); 1 / rtf = %.0f x real time" % (audio_s / wall),
                device_time_s=round(float(t_ms[-1]) * 1e-3, 3),
                frames_per_s_first_second=round(float(frames[first].sum()) / (float(t_ms[first][-1]) * 1e-3), 1) if first.any() else None,
                frames_per_s_last_second=round(float(frames[last].sum()) / ((float(t_ms[-1]) - float(t_ms[~last][-1] if (~last).any() else 0.0)) * 1e-3), 1),
                ms_per_step_first_20=round(float(t_ms[19] / 20.0), 4), ms_per_step_last_20=round(float((t_ms[-2] - t_ms[-22]) / 20.0), 4),
                shader_clock_GHz=dict(first_second=ghz(0, 1), last_second=ghz(2, 3), whole_epoch=ghz(0, 3), whole_epoch_per_xcd=ghz_xcd(0, 3),
                                      how="s_memtime ticks / s_memrealtime ticks (100 MHz) between two samples in stream order, median over the eight XCDs"),
                hwmon=hw.report("the epoch"), ingest=ing.report(), epoch_reduce=dict(collectives=1, bytes=job.red.nbytes()),
                workload=WORKLOAD_NAMES["pipeline"](a), contract=args.contract, precision=args.precision)


def decoder_facing(ctx, args, rank):
    """PCIe-inclusive rates of the NN scorer as a decoder sees them (never `value`): one buffer fill of 256 frames scored into a resident
    [256 x 10000] block (AmxHost::BatchFeatureScorer::scoreResident), then (a) nothing fetched, (b) every frame's full 40 kB row copied
    to the host (ContextScorer::score(e) on every frame), (c) 500 emissions per frame gathered (ContextScorer::scores(list))."""
    import torch

    import rasr_amd
    from tests import synth
    dims = [440] + [2048] * 6 + [10000]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision=args.precision, tuning=args.nn_tuning)
    T = 256
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(60 + rank)).standard_normal((T, 440)).astype(np.float32)).cuda()
    scores = torch.empty((T, 10000), dtype=torch.float32, device="cuda")
    host = torch.empty((T, 10000), dtype=torch.float32).pin_memory()
    rng = np.random.Generator(np.random.PCG64(61))
    rows = np.repeat(np.arange(T, dtype=np.uint32), 500)
    cols = rng.integers(0, 10000, T * 500).astype(np.uint32)
    out = {}
    for name in ("resident", "full_rows", "active_set_500"):
        def one():
            nn.score_dev(x, 440, T, scores)
            if name == "full_rows":
                host.copy_(scores, non_blocking=True)
            elif name == "active_set_500":
                ctx.gather_scores(scores, 10000, rows, cols)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100 if name == "resident" else 20
        for _ in range(n):
            one()
        torch.cuda.synchronize()
        out[name + "_frames_per_s"] = round(T * n / (time.perf_counter() - t0), 1)
    if args.precision == "f16mx" and "ksplit" not in (args.nn_tuning or ""):
        # the opt-in mode for a decoder's fixed buffer size: split-K across workgroups for passes of at most 256 frames (another order of
        # summation than the default's: scores differ by f32 rounding, tests/test_ffnn_f16mx_gpu.py)
        nk = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision=args.precision,
                                           tuning=",".join(i for i in (args.nn_tuning, "ksplit=4") if i))
        for _ in range(3):
            nk.score_dev(x, 440, T, scores)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            nk.score_dev(x, 440, T, scores)
        torch.cuda.synchronize()
        out["resident_frames_per_s_tuning_ksplit4"] = round(T * 100 / (time.perf_counter() - t0), 1)
    out["batch"] = T
    out["note"] = "cfg-4 network, one 256-frame buffer fill per pass; full rows = 40 kB per frame over the host link"
    return out


def apply_contract(args):
    """--contract -> the context's arithmetic (amx_set_contract, main()), the GMM scorers' tuning string and the oracle's mode"""
    from_tuning = [i.split("=", 1)[1] for i in (args.gmm_tuning or "").split(",") if i.strip().startswith("contract=")]
    if from_tuning:
        args.contract = from_tuning[-1]
    elif args.contract == "fma":
        args.gmm_tuning = ",".join(i for i in (args.gmm_tuning, "contract=fma") if i)


CONTRACT_TEXT = {"fma": "fma: bit-identical to the reference's default build (-march=native, GCC -ffp-contract=fast: the GMM distance accumulates "
                        "with fused multiply-adds), checked against oracle/liboracle_fma.so",
                 "off": "off: bit-identical to the reference configured with -DMARCH=x86-64 (every f32 operation rounds once), checked against "
                        "oracle/liboracle.so"}


def main():
    args = parse()
    apply_contract(args)
    launch_ranks(args)   # --gpus N without a launcher: this process becomes the launcher and exits with the ranks' status
    gpu = args.backend == "nccl"
    if not gpu and args.workload != "null":
        raise SystemExit("bench.py: --backend gloo runs the host-only stand-in only (--workload null); the product has no CPU path")
    rank, world, local = dist_setup(args)
    if not gpu:
        job = make_job(None, args, rank, world)
        dt = measure(None, job, args, world)
        import torch
        if job.walker is not None:
            job.gather_walks()
        t = torch.tensor([dt], dtype=torch.float64)
        if _dist_on():
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank == 0:
            units = job.units * args.steps * world   # weak scaling: every rank owns a batch of the same shape
            print(json.dumps({"metric": "stand-in frames/s (host-only launcher check, not the product)", "value": round(units / dt, 1),
                              "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "none", "data": "synthetic",
                              "config": {"workload": "null (no GPU work)", "backend": "gloo", "ingest": ingest_mode(args, world)},
                              "epoch_reduce": dict(collectives=1, bytes=job.red.nbytes(), backend="gloo (CPU stand-in)"),
                              "stages": job.stage_report()}))
        if _dist_on():
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    import torch

    import rasr_amd
    ctx = rasr_amd.Context(local)
    ctx.set_contract(args.contract)   # every handle and every *_dev entry point of the run follows the build of the reference the line names
    stream = torch.cuda.Stream(device=local)
    with torch.cuda.stream(stream):
        ctx.use_torch_stream()
        comm = make_comm(ctx, rank, world)
        job = make_job(ctx, args, rank, world)
        job.comm = comm
        dt = measure(ctx, job, args, world)
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if _dist_on():
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    units = job.units * args.steps * world
    if rank == 0:
        value = units / dt
        line = {"metric": "acoustic frames scored/sec (1e4-state AM)", "value": round(value, 1), "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": {"bf16": "bf16", "bf16x3": "bf16x3 (split bf16, three MFMA products per f32 product, f32 accumulate)", "fp32": "f32",
                          "f16mx": "f16+mxfp6 (per f32 product: one f16 MFMA product + one block-scaled fp6 x fp6 MFMA product for both cross "
                                   "terms, f32 accumulate)"}[args.precision]
                         if args.workload in ("pipeline", "nn-pipeline", "nn") else "f32",
                "data": "synthetic", "config": {"workload": WORKLOAD_NAMES[args.workload](args), "frames_per_step_per_gpu": job.units,
                                                "contract": CONTRACT_TEXT[args.contract]},
                "rtf": round(dt / (units * 0.01), 8), "build": rasr_amd.version()}
        if getattr(job, "requested_precision", None) and job.requested_precision != getattr(job, "nn_precision", job.requested_precision):
            line["config"]["precision_fallback"] = ("requested %s, the handle computes in %s (amx_ffnn_precision: block-maximum statistic of the weights "
                                                    "above 4.0, tuning mx_fallback=auto)" % (job.requested_precision, job.nn_precision))
        line["roofline"] = job.roofline()
        if line["roofline"] and line["roofline"].get("traffic") is not None:
            line["roofline"]["traffic_source"] = TRAFFIC_SOURCE + " (offline rocprofv3 --pmc passes on the profiling box, not this run)"
        line["stages"] = job.stage_report()
        if world == 1:
            try:
                with torch.cuda.stream(stream):
                    smi_stretch(job)
            except Exception:  # never take the headline down
                pass
        if line["roofline"] and (SHADER_CLOCK_GHZ or HWMON):
            r = line["roofline"]
            # the driver's figure where it exists; the counters' otherwise (s_memtime is a counter per CU, the CUs' counters are not
            # aligned: between two short samples on different CUs of an XCD the difference is off by their offset -- single XCDs read 1.5
            # or 2.9 GHz over 160 ms, one even ran backwards; over the seconds of the full-epoch run the offset is 0.1 %)
            sclk = HWMON["sclk_GHz"] if HWMON else SHADER_CLOCK_GHZ
            r["shader_clock_GHz"] = sclk
            if HWMON:
                r["hwmon"] = HWMON
                if HWMON.get("socket_power_W"):
                    # the step priced in joules: socket power over seconds of the same steps x the timed step (profiles/r06/energy.json has
                    # the variants -- XCD super-tile shapes, best-density formats, plain stores: none lowers the time at the 1400 W cap)
                    r["joules_per_step"] = round(HWMON["socket_power_W"] * (1e3 * dt / args.steps) * 1e-3, 3)
            # per-XCD ratios outside 0.5-1.2 x the driver's figure (without one: outside 0.5-2.45 GHz, the part's range) are the
            # unaligned-counter artefact, not a clock: masked (None), and the median is taken over what is left
            lo_, hi_ = (0.5 * HWMON["sclk_GHz"], 1.2 * HWMON["sclk_GHz"]) if HWMON else (0.5, 2.45)
            kept = [v if (v is not None and lo_ <= v <= hi_) else None for v in (SHADER_CLOCK_XCD or [])]
            vals = [v for v in kept if v is not None]
            r["shader_clock_GHz_s_memtime"] = dict(median_of_xcds=round(float(np.median(vals)), 3) if vals else None, per_xcd=kept,
                                                   masked=sum(1 for v, k in zip(SHADER_CLOCK_XCD or [], kept) if v is not None and k is None),
                                                   note="s_memtime is a per-CU counter; samples on different CUs differ by the CUs' offset -- ratios outside the plausible band are masked")
            SHADER = sclk
            if r.get("bound") == "mfma":
                r["peak_at_shader_clock"] = round(r["peak"] * SHADER / 2.4, 1)
                r["frac_at_shader_clock"] = round(r["achieved"] / (r["peak"] * SHADER / 2.4), 4)
            r["clock_note"] = ("mean sclk as the driver reports it (rocm-smi, over seconds of the same steps; the s_memtime / s_memrealtime deltas of the per-launch pass beside it).  `peak` is priced at the 2.4 GHz ceiling; the NN "
                               "GEMMs run at the package's 1400 W power cap, where the firmware holds sclk at 1.85-2.1 GHz whatever the kernel "
                               "does per cycle (profiles/r05/power_probe.log: rocm-smi power and sclk sampled during each workload)")
        if getattr(job, "ingest", None) is not None:
            line["ingest"] = job.ingest.report()
        elif hasattr(job, "setup_ingest") and getattr(job, "pcm", None) is not None and world == 1:
            line["ingest"] = dict(mode="resident", sample_format="f32", bytes_over_link_per_step=0,
                                  note="the same %d utterances every step, in HBM before the timed region" % args.utterances)
            try:
                with torch.cuda.stream(stream):
                    line["ingest"]["streamed"] = streamed_comparison(ctx, job, args, value)
            except Exception as e:  # never take the headline down
                line["ingest"]["streamed"] = dict(error=str(e)[:200])
        if hasattr(job, "red"):
            ms_ar, n_ar = ctx.profile_get("all_reduce")
            line["epoch_reduce"] = dict(collectives=1, bytes=job.red.nbytes(),
                                        backend="rccl via amx_comm_all_reduce_f64_dev (%d ranks)" % comm.world if comm else "none (single process)")
            if hasattr(job, "reduced_frames"):
                line["epoch_reduce"].update(reduced_frames=job.reduced_frames, expected_frames=job.expected_frames,
                                            reduce_ok=bool(job.reduced_frames == job.expected_frames))
        line["config"]["timing"] = ("value / ms_per_step: K steps with the library's per-launch events OFF; roofline / stages: a second pass over "
                                    "min(K, 20) steps with a HIP-event pair around every launch on the stream the kernels run on")
        if is_graph_mode(args):
            line["config"]["launch"] = "forward pass replayed as one HIP graph; roofline / stages timed in the second pass with plain launches"
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(args.workload, args.contract)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu"] = round(value / world / cb["value"], 1)
            if args.workload in ("pipeline", "nn-pipeline", "gmm-train"):
                try:
                    with torch.cuda.stream(stream):
                        ctx.use_torch_stream()
                        line["parity_check"] = parity_check(ctx, job, args)
                except Exception as e:  # never take the headline down
                    line["parity_check"] = dict(error=str(e)[:300])
        if args.workload == "pipeline" and world == 1 and not args.no_configs:
            job = None
            torch.cuda.empty_cache()
            with torch.cuda.stream(stream):
                ctx.use_torch_stream()
                line["configs"] = secondary_configs(ctx, args, rank)
                try:
                    line["decoder_facing"] = decoder_facing(ctx, args, rank)
                except Exception as e:  # never take the headline down
                    line["decoder_facing"] = dict(error=str(e)[:200])
        try:   # RCCL prints a version banner through C stdio at its own pace: push it out first, so that the JSON line is the LAST line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if _dist_on():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
