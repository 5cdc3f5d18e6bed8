#!/usr/bin/env python3
"""bench.py -- throughput of the fused HRTF convolution hot path on MI355X.

Workload (BASELINE.json configs[2], "1024 concurrent moving sources batched,
256-sample blocks, 1xMI355X"): every GPU holds 1024 looped noise sources whose azimuth
advances 1 degree per block (a crossfade of two 4-point interpolations nearly every
block).  One STEP = one call of jf_batch_run over BLOCKS_PER_STEP consecutive audio
blocks of all the rank's sources (prep -> fused FFT/multiply/IFFT/crossfade -> mix),
plus, for N > 1, the RCCL sum of the stereo mixes to rank 0.  Signals, HRTF table and
trajectories are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

prints ONE JSON line (rank 0).  metric = source-frames per second (sources x frames/s),
whole job.  `roofline` prices the fused kernel against HBM peak with ALGORITHMIC bytes
(SURVEY.md 8d); `cpu_baseline` is the float32 C oracle (oracle/, kind "port") timed on
this node's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B = 256
SOURCES_PER_GPU = 1024
BLOCKS_PER_STEP = int(os.environ.get("JF_BLOCKS_PER_STEP", "64"))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_share():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (a GPU box hands each GPU user a share of the node; more threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(jf, wl, hrir, n_sources, n_blocks):
    """The oracle (CPU restatement of the reference's path) on the first n_blocks blocks of
    the first n_sources sources of the same workload, all host threads, parallel over sources."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    ora = oracle_lib.Engine(B, 512, n_sources, hrir)
    for s in range(n_sources):
        ora.set_signal(s, wl.source_signal_and_start(s)[0])
    pos = wl.trajectories(jf, np.arange(n_sources), n_blocks)
    threads = min(oracle_lib.lib().jfo_num_threads(), cpu_share())
    ora.process_batch(pos[:2], n_threads=threads)  # warm the thread pool
    for s in range(n_sources):
        ora.reset(s)
    t0 = time.perf_counter()
    ora.process_batch(pos, n_threads=threads)
    dt = time.perf_counter() - t0
    ora.close()
    return {"value": n_sources * n_blocks * B / dt, "unit": "source-frames/s", "cores": threads,
            "kind": "port",
            "sample": f"{n_sources} sources x {n_blocks} blocks of the same moving-source workload, "
                      f"{dt:.2f} s wall on {threads} threads = the host-CPU share of this process "
                      f"(float32 C oracle, OpenMP over sources)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 256 warm-up steps (80 ms) because the first ~100 steps after an idle GPU run 10-15 % slower
    # (clock ramp; profiles/r01_experiments.md), then 512 timed steps = 32 768 blocks x 1024 sources
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--stationary", action="store_true", help="sources do not move (no crossfade)")
    ap.add_argument("--reverb", action="store_true",
                    help="BASELINE.json configs[4]: 256 sources, 128-sample blocks, 2 s convolution-reverb IR "
                         "(partitioned FDL convolution ahead of the spatialiser); not the default bench line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=256)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # JF_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the multi-rank path
    # on a 1-GPU box (RCCL refuses duplicate devices); the measured configuration is always nccl.
    backend = os.environ.get("JF_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from jf_load import jf
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "jf_workload", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
    wl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wl)

    gold = os.path.join(ROOT, "tests", "golden", "kemar_hrir_710x2x128_i16.npy")
    hrir = np.load(gold).astype(np.float32) / np.float32(32768.0)

    global B
    S = SOURCES_PER_GPU
    K, W, KB = args.steps, args.warmup, BLOCKS_PER_STEP
    ir = None
    if args.reverb:
        B, S, KB = 128, 256, 32
        rng = np.random.default_rng(99)  # exponentially decaying noise, seed 99, 2.0 s
        ir = rng.standard_normal(88200) * np.exp(-6.9 * np.arange(88200) / 88200.0)
        ir = (ir / np.sqrt((ir ** 2).sum())).astype(np.float32)
    src_lo = rank * S  # weak scaling: every rank brings its own 1024 sources
    src_ids = np.arange(src_lo, src_lo + S)

    eng = jf.Engine(B, 512, S, hrir=hrir, device=local_rank, max_batch_blocks=KB)
    for s, sid in enumerate(src_ids):
        eng.set_signal(s, wl.source_signal_and_start(sid)[0])
    if os.environ.get("JF_SOURCE_GROUP"):
        eng.set_source_group(int(os.environ["JF_SOURCE_GROUP"]))  # tuning runs only
    if ir is not None:
        eng.set_reverb(ir, 0.5)
    # The trajectories are periodic (azimuth + 1 degree per block: 360 blocks), so a long run walks one
    # uploaded period again and again instead of holding (steps x blocks x sources) records: any --steps
    # costs the same 20 B x sources x lcm(360, blocks per step) of host and device memory.
    period = int(np.lcm(360, KB))
    n_pos = period
    pos = wl.trajectories(jf, src_ids, n_pos, moving=not args.stationary)
    eng.upload_positions(pos)

    # The mix lands in a torch tensor so that RCCL can reduce it in place.  Two buffers: the
    # (latency-bound, K * 2 KB) reduce of step i runs on RCCL's stream while the engine's stream
    # already computes step i + 1; a buffer is reused only after its reduce has completed.
    mixes = [torch.zeros((KB, 2 * B), dtype=torch.float32, device="cuda") for _ in range(2)]
    pending = [None, None]
    ext = torch.cuda.ExternalStream(eng.stream_ptr())

    def step(i):
        j = i & 1
        if pending[j] is not None:
            with torch.cuda.stream(ext):
                pending[j].wait()  # stream-level wait: the engine stream must not overwrite mixes[j] early
            pending[j] = None
        eng.batch_run((i * KB) % n_pos, KB, mixes[j].data_ptr())
        if world > 1:
            with torch.cuda.stream(ext):  # the collective is ordered after the kernels just enqueued
                if backend == "nccl":
                    pending[j] = dist.reduce(mixes[j], dst=0, op=dist.ReduceOp.SUM, async_op=True)
                else:
                    pending[j] = dist.all_reduce(mixes[j], op=dist.ReduceOp.SUM, async_op=True)  # gloo: no GPU reduce

    def fence():
        for j in range(2):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Clock ramp: the first ~100 steps after an idle GPU run 10-15 % slower.  Whatever --warmup says, at
    # least 256 untimed steps run before the timed region; the extra ones are reported as "prewarm_steps".
    prewarm = max(0, 256 - W)
    for i in range(prewarm):
        step(i)
    fence()
    for i in range(W):
        step(prewarm + i)
    fence()
    if not os.environ.get("JF_NO_EVENTS"):  # tuning runs: how much do the event records cost?
        eng.profile_enable(2 if ir is not None else 1)  # level 1: two events around the fused kernel
    t0 = time.perf_counter()
    for i in range(prewarm + W, prewarm + W + K):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    reverb_ms = eng.profile_read_reverb() if ir is not None else 0.0
    eng.profile_enable(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        frames = world * S * KB * K * B
        value = frames / dt
        # algorithmic bytes of the timed windows of THIS rank (every rank has the same mix of cases)
        # the run walks the uploaded period cyclically: price each step of the period once (its
        # predecessor block is the one before it on the circle) and count how often each was timed
        abytes, rows, items = wl.algorithmic_bytes_cyclic(jf, pos, KB, B, prewarm + W, K)
        fused_s = prof["fused_ms"] * 1e-3
        achieved = abytes / fused_s / 1e9 if fused_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and ir is None and not args.stationary:  # measured for this workload only
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # what actually bounds the kernel, from the committed PMC pass of this workload:
        # VALU instructions per source-block and the share of the kernel's duration they occupy at 4 cycles each
        valu = None
        ppath = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(ppath) and ir is None and not args.stationary:
            try:
                f = json.load(open(ppath))["fused"]
                valu = {"valu_insts_per_source_block": f["SQ_INSTS_VALU"] / (S * KB),
                        "valu_issue_share_of_kernel_time": f["SQ_ACTIVE_INST_VALU"] / 1024 * 4 / (f["GRBM_GUI_ACTIVE"] / 8),
                        "source": "profiles/r01_pmc_summary.json (rocprofv3 --pmc, 1024 SIMDs, 8 XCDs)"}
            except Exception:
                valu = None
        out = {
            "metric": "source-frames/s (sources x frames/sec) at 256-sample blocks",
            "value": value, "unit": "source-frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "prewarm_steps": prewarm,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: 1024 concurrent moving sources per GPU, 256-sample blocks, "
                                   "N=1024 overlap-save, KEMAR 710x2 table"
                                   + (" (stationary variant)" if args.stationary else ""),
                       "sources_per_gpu": S, "block": B, "blocks_per_step": KB,
                       "parallelism": f"sources sharded x{world}, RCCL reduce of the stereo mix"
                       if world > 1 else "1 GPU"},
            "real_time_factor": (KB * K * B / 44100.0) / dt,
            "us_per_source_block": dt / (S * KB * K) * 1e6,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "fused_group_kernel<4>",
                         "algorithmic_bytes_per_launch": abytes / prof["launches"] if prof["launches"] else None,
                         "avg_launch_ms": prof["fused_ms"] / prof["launches"] if prof["launches"] else None,
                         "table_rows_per_source_block": rows / items,
                         "other_kernels": "prep_kernel ~9 us, mix_kernel ~5 us per launch "
                                          "(profiles/r01_kernel_stats.csv)",
                         "issue_bound": valu},
        }
        if world > 1:
            # the only exchange of the path: the sum of the per-rank stereo mixes
            out["comm"] = {"collective": "reduce(sum, dst=0) of float32[%d][%d] per step" % (KB, 2 * B),
                           "payload_bytes_per_rank_per_step": KB * 2 * B * 4,
                           "overlap": "asynchronous on RCCL's stream, double-buffered: step i + 1 computes while "
                                      "step i reduces; no collective on the data path of the kernels"}
        if ir is not None:
            # per source-block 690*129*8 B of delay line read + 129*8 B written, and the
            # 690*129*8 B of IR spectra once per block (shared by all sources)
            P = -(-len(ir) // B)
            rb = S * KB * (P * (B + 1) * 8 + (B + 1) * 8) + KB * P * (B + 1) * 8
            t = reverb_ms / max(prof["launches"], 1) * 1e-3
            out["config"]["workload"] = ("configs[4]: 256 sources + 2 s convolution-reverb IR, partitioned "
                                         "overlap-save (690 partitions of 128), 128-sample blocks")
            out["reverb_roofline"] = {"bound": "hbm", "achieved": rb / t / 1e9 if t > 0 else 0.0,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": rb / t / 1e9 / HBM_PEAK_GBS if t > 0 else 0.0, "traffic": None,
                                      "kernel": "reverb_fft_kernel + reverb_mac_tiled_kernel",
                                      "algorithmic_bytes_per_launch": rb, "avg_launch_ms": t * 1e3,
                                      # the block-tiled form reads each delay-line slot once per tile of 16
                                      # blocks and keeps the delay line (S*P KB) in the Infinity Cache, so the
                                      # per-source-block figure above is what it AVOIDS reading; what one
                                      # launch must move through HBM at least: the delay line once, the IR
                                      # spectra once, the new slots and wet blocks written
                                      "min_hbm_bytes_per_launch": S * P * B * 8 + P * B * 8 + S * KB * (B * 8 + B * 4),
                                      "multiply_accumulates_per_launch": S * KB * P * B,
                                      "note": "frac > 1 = delay-line reuse across the blocks of a tile; the kernel "
                                              "is bound by L2->L1 load bandwidth and packed-f32 FMA issue "
                                              "(DESIGN.md section 7)"}
        if world == 1 and not args.no_cpu_baseline and ir is None:
            out["cpu_baseline"] = cpu_baseline(jf, wl, hrir, S, args.cpu_sample_blocks)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
