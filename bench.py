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
whole job.  With --gpus N > 1 and no WORLD_SIZE in the environment it starts its own N
ranks (python -m torch.distributed.run) as a child process before anything touches the GPU.

`roofline`: the fused kernel is bound by fp32 vector issue, not by HBM (the 5.8 MB table is
cache-resident), so `frac` is executed fp32 flops / kernel time / 157.3 TFLOP/s; the HBM side
is reported next to it from counters collected on THIS box (two short rocprofv3 --pmc passes of
this script, run as child processes before the parent touches the GPU), or, if that is not
possible, from the committed profile with its source named.  `cpu_baseline` is the float32 C
oracle (oracle/, kind "port") timed on this node's host cores on a bounded sample of the same
workload; the same oracle run is the checker of `verified`: the last timed step's mix is compared
with it.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B = 256
SOURCES_PER_GPU = 1024
# blocks per jf_batch_run: 128 = 0.74 s of audio per launch (64: -5.5 %, 256: +3 %; profiles/r02_experiments.md)
BLOCKS_PER_STEP = int(os.environ.get("JF_BLOCKS_PER_STEP", "128"))
# --reverb (batch form): 256 blocks of 128 = the same 0.74 s of audio per launch (32: -35 %, 64: -21 %, 128: -8 %)
REVERB_BLOCKS_PER_STEP = int(os.environ.get("JF_REVERB_BLOCKS_PER_STEP", "256"))
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth (MI355X_MICROARCH.md, L2): 64 B per clock and CU at the vector L1s
HBM_ACHIEVABLE_GBS = 6300.0  # what a streaming read achieves from HBM (same guide)
FP32_VECTOR_PEAK_TF = 157.3  # MI355X fp32 vector peak (MI355X_MICROARCH.md; needs packed FMAs: 2 x 78.6)
TOL32 = 4e-7                # HIP vs float32 oracle, per source (tests/)
RV_GAIN = 0.5               # --reverb: gain of the wet signal


def kN_BLOCKS(block):
    """blocks a 1024-sample window reaches back"""
    return -(-1024 // block)


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


def load_workload():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "jf_workload", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
    wl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wl)
    return wl


def cpu_baseline_and_check(jf, wl, hrir, src_ids, pos, n_pos, last_first_block, KB, n_blocks, gpu_mix, gpu_groups, G, order,
                           reverb=None):
    """The oracle (CPU restatement of the reference's path) on the n_blocks blocks that END with the last timed
    step of the GPU run, all host threads, parallel over sources: its wall time is the CPU baseline, its output
    the check of that step (`verified`).  The oracle starts n_blocks - KB blocks earlier with empty windows; a
    window holds 1024 samples = 4 blocks, so the compared blocks see exactly the GPU's history.
    gpu_mix [KB][2B]: the GPU's mix of the last step; gpu_groups: {group index: [KB][2B]} stereo blocks of
    sampled groups of G sources, group g = sources order[G g .. G g + G - 1] (the engine's processing order; None at
    N > 1, where rank 0 only holds the reduced mix).
    reverb = (ir, gain): the convolution-reverb stage ahead of the spatialiser (oracle: jfo_reverb_set_ir, the stream
    form of cudaPart.cu:65-205); the caller then asks for P + 4 more blocks of history, P = partitions of the IR."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    S = len(src_ids)
    first = last_first_block + KB - n_blocks          # absolute block index since the start of the run
    if first < 0:   # a short run: the GPU's own history starts at block 0 with empty windows, and so does the oracle's
        n_blocks, first = n_blocks + first, 0
    ora = oracle_lib.Engine(B, 512, S, hrir)
    for j, sid in enumerate(src_ids):
        sig = wl.source_signal_and_start(sid)[0]
        ora.set_signal(j, np.roll(sig, -((first * B) % len(sig))))   # the looped stream as the GPU reads it at `first`
    if reverb is not None:
        ora.set_reverb(reverb[0], reverb[1])
    if n_pos is None:   # pos holds exactly the blocks first .. first + n_blocks - 1
        p = np.ascontiguousarray(pos[-n_blocks:])
    else:
        p = np.ascontiguousarray(pos[[(first + b) % n_pos for b in range(n_blocks)]])
    # every host core this process may use -- whatever OMP_NUM_THREADS says (torch.distributed.run exports 1 to its
    # ranks): the oracle takes the count as its num_threads clause
    threads = cpu_share()
    warm = oracle_lib.Engine(B, 512, min(S, 64), hrir)              # warm the thread pool on something else
    warm.process_batch(np.ascontiguousarray(p[:2, :min(S, 64)]), n_threads=threads)
    warm.close()
    t0 = time.perf_counter()
    omix, opart = ora.process_batch(p, want_partial=True, n_threads=threads)
    dt = time.perf_counter() - t0
    ora.close()
    what = "moving-source workload" if reverb is None else \
        f"workload (reverb stage of {-(-len(reverb[0]) // B)} partitions + spatialiser)"
    base = {"value": S * n_blocks * B / dt, "unit": "source-frames/s", "cores": threads, "kind": "port",
            "sample": f"{S} sources x {n_blocks} blocks of the same {what} (the blocks that end with "
                      f"the last timed step), {dt:.2f} s wall on {threads} threads = the host-CPU share of this "
                      f"process (float32 C oracle, OpenMP over sources)"}
    # ---- check of the last timed step
    want_mix = opart[:, -KB:].astype(np.float64).sum(axis=0)
    err_mix = float(np.abs(gpu_mix - want_mix).max())
    peak = float(np.abs(want_mix).max())
    # float32 accumulation of S sources on both sides: the tests' bound for |mix| ~ 10 is 3e-5 at S = 1024.  With the
    # reverb stage every source's signal carries the float32 sum over P partitions on both sides as well: the tests'
    # per-source bound is (2e-7 + 1e-7 sqrt(P)) max(1, |y|); the S sources' errors add like noise
    if reverb is None:
        bound = 3e-6 * max(1.0, peak) * max(1.0, float(np.sqrt(S / 1024.0)))
        tol_src = TOL32
    else:
        # (measured on MI355X at 690 partitions: 8e-8 per source, 4e-7 per group of 16, 9e-7 on the mix of 256)
        tol_src = TOL32 + 2e-8 * float(np.sqrt(-(-len(reverb[0]) // B)))
        src_peak = float(np.abs(opart[:, -KB:]).max())
        bound = tol_src * max(1.0, src_peak) * float(np.sqrt(S)) + 3e-6 * max(1.0, peak)
    ok = err_mix <= bound
    check = {"max_abs_err_mix": err_mix, "mix_peak": peak, "bound_mix": bound, "blocks_checked": KB, "sources_checked": S,
             "against": "float32 C oracle (oracle/jf_oracle.c), summed in float64"}
    if gpu_groups:
        worst = 0.0
        for g, blk in gpu_groups.items():
            want = opart[order[g * G:(g + 1) * G], -KB:].astype(np.float64).sum(axis=0)
            worst = max(worst, float(np.abs(blk - want).max()))
        check["max_abs_err_group_blocks"] = worst
        check["groups_checked"] = sorted(gpu_groups)
        # each source is held to tol_src per sample (tests/test_gpu_pair_per_source.py holds the pair kernel itself to it, one
        # live source per unit); the G sources' errors are independent and add like sqrt(G)
        check["bound_group_blocks"] = tol_src * max(1.0, 0.75 * float(np.sqrt(G)))   # tests/conftest.py: sum_tol
        ok = ok and worst <= check["bound_group_blocks"]
    return base, bool(ok), check


# ------------------------------------------------------------------------------- PMC passes --
PMC_PASSES = (("FETCH_SIZE",),
              ("WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
               "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_FLOPS_FP32"))


def collect_pmc(extra_args, want_kernels):
    """Hardware counters of the fused kernel on THIS box: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not
    fit one pass; counters only, no other trace domain besides --kernel-trace) over a short child run of this
    script.  Must run before this process touches the GPU.  Returns ({counter: mean per launch}, note)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="jf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, counters in enumerate(PMC_PASSES):
            d = os.path.join(tmp, f"p{i}")
            cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", *extra_args]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            if r.returncode != 0:
                return None, f"rocprofv3 pass {i} exited with {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if not any(k in row.get("Kernel_Name", "") for k in want_kernels):
                        continue
                    a = acc.setdefault(row["Counter_Name"], [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
            if not acc:
                return None, f"pass {i}: no counter rows for {want_kernels}"
            for c, (s, n) in acc.items():
                out[c] = s / n
    except Exception as ex:  # a profiler problem must not cost the bench line
        return None, f"{type(ex).__name__}: {ex}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, "rocprofv3 --kernel-trace --pmc, two passes of `bench.py --pmc-child` on this box"


def single_source_latency_us(jf, hrir):
    """configs[1]: one source, one 256-sample block per call through jf_process_block (north_star: <= 300 us)."""
    e = jf.Engine(256, 512, 1, hrir=hrir)
    rng = np.random.default_rng(7)
    e.set_signal(0, rng.uniform(-0.5, 0.5, 44100).astype(np.float32))
    # the C entry point itself on a buffer allocated once, as a C host calls it (the Python wrapper's fresh output array
    # and status check cost ~3 us per call, which is not the library's)
    out = np.zeros(512, np.float32)
    fp, call, h = out.ctypes.data_as(jf._f), jf.lib().jf_process_block, e.h
    t = []
    for i in range(400):
        e.set_spherical(0, 5.0, float(i % 360), 0.5)  # moving: a crossfade every block
        t0 = time.perf_counter()
        rc = call(h, fp)
        t.append(time.perf_counter() - t0)
        if rc != 0:
            raise RuntimeError(f"jf_process_block returned {rc}")
    e.close()
    t = np.array(t[100:]) * 1e6
    return {"median": float(np.median(t)), "p99": float(np.percentile(t, 99)), "calls": len(t),
            "what": "jf_process_block (the C entry point, output buffer allocated once), 1 moving source -- a crossfade "
                    "every block --, B = 256 (configs[1]); host call to host result"}


def realtime_reverb_call_us(jf, hrir, S, B, ir, gain):
    """configs[4] through the real-time ENTRY POINT: one block per jf_process_block call, host call to host result, calls back
    to back -- the one-launch spatialiser behind the reverb stage's head kernel, the big partitions' work on the engine's
    second stream.  (The timed region above goes through jf_batch_run with one block per call and HIP events round the
    stage: a pipelined, profiled run, in which the engine keeps everything in line on one stream.)"""
    e = jf.Engine(B, 512, S, hrir=hrir)
    for s in range(S):
        e.set_signal(s, np.random.default_rng(1234 + s).uniform(-0.5, 0.5, 44100).astype(np.float32))
        e.set_spherical(s, -40 + (s * 7) % 121, (s * 37) % 360, 1.0)
    e.set_reverb(ir, gain)
    out = np.zeros(2 * B, np.float32)
    fp, call, h = out.ctypes.data_as(jf._f), jf.lib().jf_process_block, e.h
    t = []
    for k in range(64 + 1600):
        if k % 7 == 0:  # a fifth of the sources move now and then, as they would
            for s in range(0, S, 5):
                e.set_spherical(s, -40 + (s * 7) % 121, (s * 37 + k) % 360, 1.0)
        t0 = time.perf_counter()
        rc = call(h, fp)
        t.append(time.perf_counter() - t0)
        if rc != 0:
            raise RuntimeError(f"jf_process_block returned {rc}")
    side = any(k.endswith("@side") for k in e.last_kernels())  # call 1664 = the last block of a big block of 16
    e.close()
    t = np.array(t[64:]) * 1e6
    return {"mean": float(t.mean()), "median": float(np.median(t)), "p99": float(np.percentile(t, 99)), "max": float(t.max()),
            "calls": len(t), "by_place_in_the_cycle_of_16_blocks_median": [float(v) for v in np.median(t.reshape(-1, 16), axis=0)]
            if B <= 128 else None, "big_partitions_on_the_side_stream": bool(side),
            "what": f"jf_process_block, {S} sources, B = {B}, {len(ir)} taps of response, calls back to back, a fifth of the "
                    "sources moves every 7th block; host call to host result"}


def spawn_ranks(n, argv):
    """--gpus N without a launcher: start the N ranks as a child (this process has not touched the GPU)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    # torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks unless the variable is set: the CPU baseline of an
    # N > 1 line must use the same host cores as the N = 1 line's (it passes its own thread count as well)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(cpu_share()))
    return subprocess.call(cmd, env=env)


def reverb_response(n_ir):
    """exponentially decaying noise, seed 99 (2.0 s = 88 200 taps for configs[4]), unit energy"""
    rng = np.random.default_rng(99)
    ir = rng.standard_normal(n_ir) * np.exp(-6.9 * np.arange(n_ir) / float(n_ir))
    return (ir / np.sqrt((ir ** 2).sum())).astype(np.float32)


def leg_pmc_kernels(args):
    """--reverb: the counters of the stage's product kernel -- batch calls: the big partitions' (or, with uniform partitions,
    the tiled kernel); one-block calls: the head's kernel, which runs every block"""
    return ("reverb_mac_kernel",) if args.realtime else ("reverb_big_mac_kernel", "reverb_mac_tiled_kernel")


def precollect_pmc(args, world):
    """The counter passes of one configuration (collect_pmc), or (None, why not)."""
    if world > 1:
        return None, ("not collected: the counter passes run only at N = 1 (a profiled child per rank would share the GPUs with "
                      "the measurement)")
    if args.pmc_child or args.no_pmc:
        return None, "not collected (--no-pmc / --pmc-child)"
    extra = ((["--stationary"] if args.stationary else []) + (["--reverb"] if args.reverb else [])
             + (["--move-every", str(args.move_every)] if args.move_every != 1 else [])
             + (["--realtime"] if args.realtime else [])
             + (["--rv-sources", str(args.rv_sources), "--rv-ir-seconds", str(args.rv_ir_seconds)] if args.reverb else []))
    return collect_pmc(extra, leg_pmc_kernels(args) if args.reverb else ("fused_pair_kernel", "fused_block_kernel"))


def also_configurations(args):
    """{name: args of the leg}: what the default line reports beside its headline -- every figure of
    BASELINE.json's other single-GPU configurations in the one record a machine the builder never touched produces."""
    import copy
    rv = copy.copy(args)
    rv.reverb, rv.steps, rv.warmup = True, max(32, args.steps), max(8, args.warmup)
    st = copy.copy(args)
    st.stationary, st.steps = True, max(20, args.steps)
    rv.also_leg = st.also_leg = True
    return {"reverb": rv, "reverb_realtime_us": None, "stationary": st}


def brief(out, keep_roofline):
    """A leg's full line cut down to what the "also" record carries."""
    if out is None:
        return None
    b = {k: out.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "prewarm_steps", "ms_per_step", "real_time_factor",
                                 "verified", "step_split_ms")}
    b["config"] = {k: out["config"].get(k) for k in ("workload", "sources_per_gpu", "block", "blocks_per_step", "source_group",
                                                     "kernels", "interp_table")}
    r = out.get("roofline") or {}
    b["roofline"] = {k: r.get(k) for k in keep_roofline if k in r}
    if out.get("cpu_baseline"):
        b["cpu_baseline"] = {k: out["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample", "gpu_over_cpu")}
    if out.get("verification"):
        b["verification"] = {k: out["verification"].get(k) for k in ("max_abs_err_mix", "mix_peak", "bound_mix", "max_abs_err_group_blocks",
                                                                     "bound_group_blocks", "against")}
    return b


def run_leg(args, ctx, pmc, pmc_note):
    """One configuration from engine creation to its result line (a dict on rank 0, None elsewhere): warm-up, the timed region
    bracketed by barriers and synchronisation, the roofline of its dominant kernel, the CPU baseline and the check of the last
    timed step against the oracle.  main() runs the headline configuration through it and then -- same process, fresh engines
    -- the secondary ones whose figures go under the line's "also" key."""
    torch, dist, jf, wl, hrir = ctx["torch"], ctx["dist"], ctx["jf"], ctx["wl"], ctx["hrir"]
    rank, local_rank, world, use_dist, backend, host_info = (ctx[k] for k in ("rank", "local_rank", "world", "use_dist", "backend", "host_info"))
    rv_pmc_kernels = leg_pmc_kernels(args)
    out = None
    global B
    B = 256
    S = SOURCES_PER_GPU
    K, W, KB = args.steps, args.warmup, BLOCKS_PER_STEP
    if args.pmc_child:
        K, W = (40, 10) if args.realtime else (6, 2)
    ir = None
    if args.reverb:
        B, S, KB = 128, args.rv_sources, (1 if args.realtime else REVERB_BLOCKS_PER_STEP)
        ir = reverb_response(int(round(args.rv_ir_seconds * 44100)))
    src_lo = rank * S  # weak scaling: every rank brings its own 1024 sources
    src_ids = np.arange(src_lo, src_lo + S)

    eng = jf.Engine(B, 512, S, hrir=hrir, device=local_rank, max_batch_blocks=KB)
    for s, sid in enumerate(src_ids):
        eng.set_signal(s, wl.source_signal_and_start(sid)[0])
    if os.environ.get("JF_SOURCE_GROUP"):
        eng.set_source_group(int(os.environ["JF_SOURCE_GROUP"]))  # tuning runs only
    # tuning runs only (profiles/r04_interp_*.sh, rt_ab.sh): the library itself reads nothing from the environment since
    # round 6 -- the A/B scripts' variables are turned into calls of jefferson_debug.h's setters here
    if os.environ.get("JF_INTERP_TABLE") in ("0", "1", "2"):
        eng.set_interp_table(int(os.environ["JF_INTERP_TABLE"]))
    if os.environ.get("JF_RV_SIDE_WGS"):
        eng.set_reverb_side_workgroups(int(os.environ["JF_RV_SIDE_WGS"]))
    if ir is not None:
        if os.environ.get("JF_RV_PARTITIONING"):   # 1 = uniform partitions (round 3's form), 2 = non-uniform; default: by length
            eng.set_reverb_partitioning(int(os.environ["JF_RV_PARTITIONING"]))
        eng.set_reverb(ir, RV_GAIN)
    # The trajectories are periodic (azimuth + 1 degree per block: 360 blocks), so a long run walks one
    # uploaded period again and again instead of holding (steps x blocks x sources) records: any --steps
    # costs the same 20 B x sources x lcm(360, blocks per step) of host and device memory.
    n_pos = int(np.lcm(360 * args.move_every, KB))
    if n_pos > 360 * 128:  # (--move-every with a long period: one pass through a shorter stretch, walked cyclically: the seam
        n_pos = KB * max(1, (360 * 128) // KB)   # is one more move)
    # tuning runs only (profiles/r04_interp_ab.sh): every source at elevation 5 -- 360 distinct positions, mostly four-row interpolations, whose
    # pre-interpolated rows (2.9 MB) stay in the caches; the line then says so in config.workload
    narrow = os.environ.get("JF_BENCH_NARROW") is not None
    pos = wl.trajectories(jf, src_ids, n_pos, moving=not args.stationary, ele_override=5 if narrow else None,
                          move_every=args.move_every)
    eng.upload_positions(pos)

    if args.pmc_child:  # a few launches for the counters, no torch, no timing
        for i in range(W + K):
            eng.batch_run((i * KB) % n_pos, KB)
        eng.synchronize()
        eng.close()
        return None

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
        if use_dist:
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
        if use_dist:
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
    stride = args.event_stride
    if not os.environ.get("JF_NO_EVENTS"):  # tuning runs: how much do the event records cost?
        eng.profile_enable(2 if ir is not None else 1)  # level 1: two events around the fused kernel
        # A pair of event records costs ~7 us of stream time -- 2.7 % of a 0.25 ms step (1.30e11 against 1.33e11 with
        # JF_NO_EVENTS=1, profiles/r03_experiments.md): the fused launch is timed at every EVENT_STRIDE-th step of the timed
        # region, and its average is over those launches ("launches_timed").  With the reverb every kernel of such a step
        # is timed (eight records).
        # A short run -- the driver's --steps 20 -- would rest on two or three timed launches at that stride: the stride is
        # shortened until at least eight launches are timed (every 2nd of 20; every one below 16 steps).  Around EVERY launch
        # the records cost 12 us of a 0.25 ms step (measured: 0.2585 against 0.2465 ms, profiles/r04/bench_driver_shape.json of
        # the first collection), which is in ms_per_step and `value`: no more of them than the average needs.
        stride = max(1, min(args.event_stride, K // 8))
        eng.profile_set_stride(stride)
    t0 = time.perf_counter()
    for i in range(prewarm + W, prewarm + W + K):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    reverb_ms = eng.profile_read_reverb() if ir is not None else 0.0
    eng.profile_enable(False)
    kernels = eng.last_kernels()
    G = eng.last_source_group()
    order = eng.source_order()

    # what the last timed step left behind (checked against the oracle below)
    i_last = prewarm + W + K - 1
    last_mix = mixes[i_last & 1].cpu().numpy().copy()
    groups = {}
    if world == 1:
        part = eng.read_device(eng.partial_device_ptr(), (KB, S // G, 2 * B))
        for g in sorted({0, 1, (S // G) // 3, (S // G) // 2, S // G - 2, S // G - 1}):
            groups[int(g)] = part[:, g].copy()

    # prep / mix kernel times: a short untimed pass with every kernel bracketed by events
    other = {}
    if ir is None:
        eng.profile_enable(2)
        for i in range(prewarm + W + K, prewarm + W + K + 16):
            step(i)
        fence()
        p2 = eng.profile_read()
        eng.profile_enable(False)
        other = {"prep_kernel_us": p2["prep_ms"] / max(p2["launches"], 1) * 1e3,
                 "mix_kernel_us": p2["mix_ms"] / max(p2["launches"], 1) * 1e3,
                 "source": "HIP events on the engine stream, 16 untimed steps after the timed region"}

    # SURVEY.md 8(d) config 4 asks for the communication fraction.  In the timed region the collective of step i overlaps
    # step i + 1's kernels, so it is measured on its own afterwards: 16 more steps in which the engine's stream waits for the
    # step's own collective, HIP events on that stream -- the first when the step's kernels are done, the second when the
    # reduced mix is there.
    comm_ms = None
    if use_dist:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(16)]
        base_i = prewarm + W + K + (16 if ir is None else 0)
        for n, (e0, e1) in enumerate(evs):
            i = base_i + n
            j = i & 1
            eng.batch_run((i * KB) % n_pos, KB, mixes[j].data_ptr())
            with torch.cuda.stream(ext):
                e0.record(ext)
                if backend == "nccl":
                    wk = dist.reduce(mixes[j], dst=0, op=dist.ReduceOp.SUM, async_op=True)
                else:
                    wk = dist.all_reduce(mixes[j], op=dist.ReduceOp.SUM, async_op=True)
                wk.wait()
                e1.record(ext)
        fence()
        comm_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        t = torch.tensor([comm_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm_ms = float(t.item())

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        frames = world * S * KB * K * B
        value = frames / dt
        # the launches that were timed (every --event-stride-th step); the workload totals below are over all K steps, so
        # the timed launches' time is scaled to K launches: fused_s = K x the average timed launch
        timed = max(prof["launches"], 1)
        launches = K
        fused_s = prof["fused_ms"] * 1e-3 / timed * K
        first_step = prewarm + W
        # the run walks the uploaded period cyclically: price each step of the period once (its predecessor block is
        # the one before it on the circle) and count how often each was timed
        pre_rows = bool(eng.last_run_used_rows())   # the timed runs read pre-interpolated rows (include/jefferson.h)
        abytes, rows, items = wl.algorithmic_bytes_cyclic(jf, pos, KB, B, first_step, K, pre_rows=pre_rows)
        pair = any("pair" in k for k in kernels)
        flops_ex, flops_ref = wl.flops_cyclic(jf, pos, KB, B, G, first_step, K, old_sets_spectral=pair, pre_rows=pre_rows)
        tf = flops_ex / fused_s / 1e12 if fused_s > 0 else 0.0
        fused_name = next((k for k in kernels if k.startswith("fused_")), kernels[-1])
        # "fused_pair_kernel<4>+prep": the launch's trailing workgroups prepare the next window's descriptors (same kernel
        # symbol in rocprofv3: fused_pair_kernel<4>); its duration and its counters include them
        fused_has_prep = fused_name.endswith("+prep")
        fused_name = fused_name.split("+")[0]
        roof = {"bound": "valu-fp32", "achieved": tf, "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / FP32_VECTOR_PEAK_TF, "traffic": None, "kernel": fused_name,
                "avg_launch_ms": prof["fused_ms"] / timed,
                "launches_timed": timed,
                "launch_timing": ("HIP events on the engine's stream around every %d-th fused launch of the timed region (%d of %d"
                                  " launches; a pair of event records costs ~7 us of stream time)" % (stride, timed, K)
                                  if ir is None else
                                  "HIP events around every kernel of every %d-th step of the timed region (%d of %d steps)"
                                  % (stride, timed, K)),
                "launch_includes": ("the next window's descriptors (index/weight rule for 131 072 items in trailing "
                                    "workgroups, ~2.5 us of the launch, ~20 of the VALU instructions per source-block)"
                                    if fused_has_prep else None),
                "flops_per_launch_executed": flops_ex / launches,
                "flop_model": "jefferson-2.0_amd/workload.py flops_window (textbook counts; tests/test_abi.py)",
                "table_rows_per_source_block": rows / items,
                "other_kernels": other,
                "why_not_hbm": "the 5.8 MB HRTF table is cache-resident: compulsory HBM bytes are ~1.3 KB per source-block",
                # what the vector unit sustains on this chip with every CU busy (profiles/micro/valu_rate.hip, 4 waves per
                # SIMD, register operands): the spec figure assumes 2.4 GHz, the clock under that load is ~1.9-2.0 GHz
                "peak_sustained_measured": {"v_fma_f32": 120.2, "v_pk_fma_f32": 137.4, "unit": "TFLOP/s",
                                            "frac_of_v_fma_f32": tf / 120.2}}
        # NOT a roofline fraction (it prices work the kernel does not do): the same output priced with the reference's own
        # algorithm -- an unpruned inverse pair per filter set and source, GPUSoundSource.cu:320-385 --, comparable across
        # kernels that skip different amounts of that work
        ref_rate = {"flops_per_launch_reference_algorithm": flops_ref / launches,
                    "tflops_if_the_reference_algorithm_were_executed": (flops_ref / fused_s / 1e12) if fused_s > 0 else 0.0,
                    "note": "work-equivalent rate for comparisons between kernel generations, not an achieved figure"}
        # SURVEY.md 8(d)'s algorithmic bytes (table rows re-read per item): a CACHE-level rate, not an HBM rate
        roof["algorithmic_cache_gbps"] = abytes / fused_s / 1e9 if fused_s > 0 else 0.0
        roof["algorithmic_bytes_per_launch"] = abytes / launches
        hbm = {"peak": HBM_PEAK_GBS, "unit": "GB/s"}
        fpmc = None if ir is not None else pmc   # with --reverb the counters are the multiply-accumulate kernel's
        # How busy the compute units' vector-memory return path is (round 5, profiles/r05/l1_bound.md): a vector L1 returns 64
        # bytes per clock (256 CUs x 64 B x ~2.1 GHz = the guide's 34.5 TB/s for the L2s, the same pipe from the other
        # side).  From the kernel's own load count -- the algorithmic figure above counts 6.65 rows per source-block, the
        # kernel loads ~4.2 (old and new filter sets mostly share their rows): 16-byte row loads of 1 KB per wave, eight
        # 8-byte window loads of 512 B.  Half busy, and not what the kernel waits for (closer rows -3..-4.5 %, more loads in
        # flight nothing): reported so that the third resource is on the line beside the vector unit and HBM.
        if fpmc and fpmc.get("SQ_INSTS_VMEM_RD") and fpmc.get("GRBM_GUI_ACTIVE") and fused_s > 0:
            loads = fpmc["SQ_INSTS_VMEM_RD"]                      # wave-instructions per launch
            win = 8.0 * S * KB                                     # the windows' (8 B per lane)
            l1_bytes = max(loads - win, 0.0) * 1024.0 + min(loads, win) * 512.0
            cyc = fpmc["GRBM_GUI_ACTIVE"] / 8.0                   # (summed over the 8 XCDs) cycles of one launch under the counters
            roof["l1"] = {"what": "bytes the vector L1s return to registers per launch, from the kernel's own load count",
                          "bytes_per_launch": l1_bytes, "achieved": l1_bytes / (fused_s / launches) / 1e9, "unit": "GB/s",
                          "peak": L2_PEAK_GBS, "peak_source": "MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s = 64 B per clock and CU",
                          "frac": l1_bytes / (fused_s / launches) / 1e9 / L2_PEAK_GBS,
                          "bytes_per_clock_and_cu": l1_bytes / cyc / 256.0,
                          "clock_ghz_under_the_counters": cyc / (fused_s / launches) / 1e9,
                          "loads_per_source_block": loads / (S * KB), "evidence": "profiles/r05/l1_bound.md", "source": pmc_note}
        if fpmc and "FETCH_SIZE" in fpmc and "WRITE_SIZE" in fpmc:
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads 1/2 of a wide
            # coalesced stream -> x2; WRITE_SIZE is exact for 16-B-per-lane stores
            traffic = fpmc["FETCH_SIZE"] * 1024 * 2 + fpmc["WRITE_SIZE"] * 1024
            src = pmc_note
        else:
            traffic, src = None, None   # never a figure from another run: null and the reason (`source`)
        if traffic is not None:
            roof["traffic"] = traffic
            hbm.update({"bytes_per_launch": traffic, "achieved": traffic / (fused_s / launches) / 1e9,
                        "frac": traffic / (fused_s / launches) / 1e9 / HBM_PEAK_GBS, "source": src})
        else:
            hbm.update({"bytes_per_launch": None, "source": f"unavailable: {pmc_note}"})
        roof["hbm"] = hbm
        if fpmc and "SQ_INSTS_VALU" in fpmc:
            iss = {"valu_insts_per_source_block": fpmc["SQ_INSTS_VALU"] / (S * KB), "source": pmc_note}
            if fpmc.get("GRBM_GUI_ACTIVE") and fpmc.get("SQ_INSTS_VALU"):
                # A SIMD retires one wave64 vector instruction per ~2 cycles when two or more of its waves have one
                # ready (MI355X_MICROARCH.md; profiles/micro/valu_rate.hip measures 1.09 ns at 4 waves per SIMD), one
                # per 4 cycles from a single wave; packed-f32 instructions take twice that.  GRBM_GUI_ACTIVE is summed
                # over the 8 XCDs, SQ_INSTS_VALU over the 1024 SIMDs.
                per_simd = fpmc["SQ_INSTS_VALU"] / 1024
                cycles = fpmc["GRBM_GUI_ACTIVE"] / 8
                iss["valu_pipe_busy_share_at_2_cycles_per_inst"] = per_simd * 2 / cycles
                iss["valu_issue_share_at_4_cycles_per_inst"] = per_simd * 4 / cycles
                iss["valu_rate_note"] = ("2 cycles: what the SIMD needs with >= 2 waves ready (plain f32; packed f32 twice "
                                         "that); 4 cycles: what one wave alone can issue")
            if fpmc.get("SQ_ACTIVE_INST_VALU") and fpmc.get("GRBM_GUI_ACTIVE"):
                # The bound the kernel actually sits on: the share of the launch in which a SIMD's vector pipe is executing a
                # vector instruction (SQ_ACTIVE_INST_VALU counts quad-cycles, summed over the 1024 SIMDs; GRBM_GUI_ACTIVE
                # cycles, summed over the 8 XCDs).  The flop fraction (`frac`) prices the same work at 2 cycles per
                # instruction; executed, an instruction of this kernel's mix occupies the pipe for ~4.
                iss["valu_active_share"] = (fpmc["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (fpmc["GRBM_GUI_ACTIVE"] / 8.0)
            if fpmc.get("SQ_INSTS_VMEM_RD"):
                iss["vmem_loads_per_source_block"] = fpmc["SQ_INSTS_VMEM_RD"] / (S * KB)
            if fpmc.get("SQ_WAVE_CYCLES") and fpmc.get("SQ_WAIT_ANY"):
                iss["wave_time_waiting_share"] = fpmc["SQ_WAIT_ANY"] / fpmc["SQ_WAVE_CYCLES"]
            roof["issue"] = iss
        if fpmc and fpmc.get("SQ_INSTS_VALU_FLOPS_FP32") and fused_s > 0:
            # the hardware's own count of executed fp32 flops (per wave instruction: x 64 lanes), beside the textbook model
            fl = fpmc["SQ_INSTS_VALU_FLOPS_FP32"] * 64.0
            roof["frac_pmc"] = fl / (fused_s / launches) / 1e12 / FP32_VECTOR_PEAK_TF
            roof["flops_per_launch_pmc"] = fl
            roof["frac_pmc_source"] = "SQ_INSTS_VALU_FLOPS_FP32 x 64 lanes per launch / avg_launch_ms; " + pmc_note
        else:
            roof["frac_pmc"] = None
        out = {
            "metric": "source-frames/s (sources x frames/sec) at 256-sample blocks",
            "value": value, "unit": "source-frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "prewarm_steps": prewarm,
            "prewarm_policy": ("whatever --warmup says, at least 256 untimed steps (~65 ms of GPU time) run before the timed "
                               "region: the first ~100 steps after an idle GPU run 10-15 % slower (clock ramp), so `value` is "
                               "a warmed steady-state figure; prewarm_steps = max(0, 256 - warmup) of them are this script's own"),
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "host": host_info,
            "config": {"workload": "configs[2]: 1024 concurrent moving sources per GPU, 256-sample blocks, "
                                   "N=1024 overlap-save, KEMAR 710x2 table"
                                   + (" (stationary variant)" if args.stationary else "")
                                   + (f" (VARIANT: the sources move every {args.move_every}-th block)" if args.move_every != 1 else "")
                                   + (" (TUNING VARIANT: every source at elevation 5)" if narrow else ""),
                       "interp_table": {"setting": ["off", "always", "per run"][eng.interp_table()],
                                        "rows_read_by_the_timed_runs": eng.last_run_used_rows()},
                       "sources_per_gpu": S, "block": B, "blocks_per_step": KB, "source_group": G,
                       "source_order": "by table row of the first position" if not np.array_equal(order, np.arange(S))
                       else "as given",
                       "kernels": kernels,
                       "parallelism": (f"sources sharded x{world}, "
                                       + ("RCCL reduce" if backend == "nccl" else f"{backend} all_reduce (rehearsal, not RCCL)")
                                       + " of the stereo mix") if use_dist else "1 GPU"},
            "real_time_factor": (KB * K * B / 44100.0) / dt,
            "us_per_source_block": dt / (S * KB * K) * 1e6,
            "roofline": roof,
            "reference_algorithm_rate": ref_rate,
        }
        if use_dist:
            # the only exchange of the path: the sum of the per-rank stereo mixes
            out["comm"] = {"collective": ("reduce(sum, dst=0)" if backend == "nccl" else "all_reduce(sum)")
                                         + " of float32[%d][%d] per step" % (KB, 2 * B),
                           "backend": "RCCL" if backend == "nccl" else backend,
                           "payload_bytes_per_rank_per_step": KB * 2 * B * 4,
                           "overlap": "asynchronous on the collective's stream, double-buffered: step i + 1 computes "
                                      "while step i reduces; no collective on the data path of the kernels",
                           # the collective's own duration against a step: what it WOULD cost if it were not overlapped
                           "ms_per_collective": comm_ms,
                           # only an RCCL collective has a share worth quoting: a rehearsal backend (gloo through the host)
                           # exercises the measurement, its duration is in ms_per_collective, and it says nothing about xGMI
                           "fraction": (comm_ms / (dt / K * 1e3)) if (comm_ms is not None and backend == "nccl") else None,
                           "fraction_null_because": None if backend == "nccl" else
                           f"the backend is {backend}, not RCCL: not a figure of the multi-GPU path",
                           "fraction_is": "duration of one collective / ms_per_step; the collective runs beside the next "
                                          "step's kernels, so this is an upper bound of its share of the step, not time added "
                                          "to it",
                           "measured": "HIP events on the engine's stream around the collective (kernels done -> reduced mix "
                                       "there), 16 steps after the timed region with the stream waiting for its own step's "
                                       "collective; max over ranks"}
        if ir is not None:
            P = -(-len(ir) // B)
            t = reverb_ms / timed * 1e-3  # average time of the reverb stage's kernels over the steps that were timed
            n_tot, p_head, p_big, big_taps = eng.reverb_partitions()
            std = S == 256 and len(ir) == 88200
            parts = ((f"{p_head} partitions of {B} + {p_big} of {big_taps}" if args.realtime else
                      f"{p_big + 2} partitions of {big_taps} for the whole big blocks of a call; {p_head} of {B} + {p_big} of {big_taps} "
                      "for blocks worked on their own") if p_big else f"{P} partitions of {B}")
            out["config"]["workload"] = (("configs[4]: " if std else "configs[4] scaled: ")
                                         + f"{S} sources + {len(ir) / 44100.0:g} s convolution-reverb IR ({P} blocks long), "
                                           f"partitioned overlap-save ({parts}), {B}-sample blocks"
                                         + (", ONE block per call (real-time shape)" if args.realtime else ""))
            rv_kernels = [k for k in kernels if k.startswith("reverb_")]
            rv = {"kernel": " + ".join(rv_kernels), "kernels_are": "the reverb stage's kernels of the LAST timed step",
                  "avg_stage_ms": t * 1e3, "stage_timing": "HIP events around the stage's kernels on the engine's stream"}
            if p_big:
                # Non-uniform partitioning (jf_device.h: ReverbBigParams).  Per big block (M blocks) and source: one
                # transform of 2 B1 samples, p_big + 2 (FULL: blocks inside a batch call -- all the big partitions) or p_big
                # (TAIL: those behind the head) spectra of the delay line against as many partition spectra, one inverse;
                # blocks worked on their own add the head's 2 M partitions of B.  The stage is bound by the delay-line stream (HBM): bytes below are ALGORITHMIC --
                # every spectrum a product needs counted once per tile of 16 products (the kernel keeps a sliding window).
                M, B1 = big_taps // B, big_taps
                if args.realtime:
                    # averages over the 16-block cycle: one transform, one TAIL product + inverse, 16 heads
                    fdl_read = S * p_big * B1 * 8 / M
                    rb = fdl_read + S * (B1 * 8 * 3 + 2 * B1 * 4) / M + S * (p_head * B * 8 + B * 8 + 2 * B * 4)
                    macs = S * (p_big * B1 / M + p_head * B)
                    flops = 8.0 * macs + S * (2 * 5 * B1 * np.log2(B1) / M + 2 * 5 * B * np.log2(B))
                    fdl_bytes = S * (p_big + 1) * B1 * 8
                else:
                    p_full = p_big + 2
                    nb_big = KB // M   # whole big blocks per call (KB is a multiple of M: the bench's calls are aligned)
                    fdl_read = S * (p_full + min(nb_big, 16) - 1) * B1 * 8 * (-(-nb_big // 16))
                    rb = fdl_read + p_full * B1 * 8 + 3 * S * nb_big * B1 * 8 + 2 * S * KB * B * 4
                    macs = S * nb_big * p_full * B1
                    flops = 8.0 * macs + 2 * S * nb_big * 5 * B1 * np.log2(B1)
                    fdl_bytes = S * (p_full + nb_big) * B1 * 8
                gbs = rb / t / 1e9 if t > 0 else 0.0
                rv.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                           "traffic": None, "algorithmic_bytes_per_step": rb, "delay_line_bytes_read_per_step": fdl_read,
                           "multiply_accumulates_per_step": macs, "flops_per_step": flops,
                           "valu_fp32_frac": flops / t / 1e12 / FP32_VECTOR_PEAK_TF if t > 0 else 0.0,
                           "delay_line_bytes_resident": fdl_bytes,
                           "level": ("the delay line (%d MB) fits the 256 MiB Infinity Cache: part of the stream comes out of it"
                                     % (fdl_bytes // 10 ** 6)) if fdl_bytes <= 256 * 2 ** 20 else "hbm (the delay line is larger "
                                    "than the 256 MiB Infinity Cache)",
                           "uniform_partitioning_would_need": {"multiply_accumulates_per_step": S * KB * P * B,
                                                               "delay_line_bytes_read_per_step": S * KB * P * B * 8 if args.realtime
                                                               else S * (-(-KB // 16)) * (P + 15) * B * 8},
                           "frac_of_achievable_6300": gbs / HBM_ACHIEVABLE_GBS})
            else:
                # uniform partitioning (per source-block P*129*8 B of delay line read + 129*8 B written, and
                # the P*129*8 B of IR spectra once per block)
                rb = S * KB * (P * (B + 1) * 8 + (B + 1) * 8) + KB * P * (B + 1) * 8
                macs = S * KB * P * B
                rv.update({"algorithmic_bytes_per_step": rb, "multiply_accumulates_per_step": macs})
                if args.realtime:
                    fdl_bytes = S * (P + KB) * B * 8
                    in_mall = fdl_bytes + P * B * 8 <= 256 * 2 ** 20
                    gbs = rb / t / 1e9 if t > 0 else 0.0
                    rv.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                               "level": "infinity-cache (the delay line fits the 256 MiB MALL: NOT an HBM figure)" if in_mall
                                        else "hbm (the delay line is larger than the 256 MiB Infinity Cache)",
                               "delay_line_bytes": fdl_bytes,
                               "frac_of_achievable_6300": gbs / HBM_ACHIEVABLE_GBS,
                               "achievable_note": "MI355X_MICROARCH.md: 8 TB/s spec peak, ~6.3 TB/s achievable from HBM"})
                else:
                    tfr = 8.0 * macs / t / 1e12 if t > 0 else 0.0
                    rv.update({"bound": "valu-fp32", "achieved": tfr, "peak": FP32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                               "frac": tfr / FP32_VECTOR_PEAK_TF, "traffic": None,
                               "algorithmic_cache_gbps": rb / t / 1e9 if t > 0 else 0.0,
                               "min_hbm_bytes_per_launch": S * P * B * 8 + P * B * 8 + S * KB * (B * 8 + B * 4)})
            if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                rv["traffic"] = pmc["FETCH_SIZE"] * 1024 * 2 + pmc["WRITE_SIZE"] * 1024
                rv["traffic_is"] = ("per launch of the stage's product kernel only (" + ", ".join(rv_pmc_kernels) + "): FETCH_SIZE x 2 + "
                                    "WRITE_SIZE; FETCH_SIZE counts Infinity-Cache hits too; " + pmc_note)
            # Which kernel dominates the step: with uniform partitions the reverb's multiply-accumulate kernel (5/6 of a step);
            # with non-uniform ones the spatialiser's fused kernel again (~40 %).  `roofline` is the reverb stage's -- this
            # configuration's subject -- and the spatialiser's fused kernel keeps its own under another key.
            out["spatialiser_roofline"] = out["roofline"]
            out["roofline"] = rv
            out["step_split_ms"] = {"reverb_stage": t * 1e3, "spatialiser_fused_kernel": prof["fused_ms"] / timed,
                                    "whole_step": dt / K * 1e3}
            out["metric"] = "source-frames/s (sources x frames/sec) at 128-sample blocks, reverb + spatialiser"
        if world == 1 and ir is None and not getattr(args, "also_leg", False):
            try:
                out["single_source_block_latency_us"] = single_source_latency_us(jf, hrir)
            except Exception as ex:
                out["single_source_block_latency_us"] = {"error": str(ex)}
        if world == 1 and ir is not None and args.realtime:
            try:
                out["realtime_call_us"] = realtime_reverb_call_us(jf, hrir, S, B, ir, RV_GAIN)
            except Exception as ex:
                out["realtime_call_us"] = {"error": str(ex)}
        if not args.no_cpu_baseline:
            all_ids = np.arange(0, world * S)
            nb = max(KB + 4, min(args.cpu_sample_blocks, max(KB + 4, 262144 // len(all_ids))))
            if ir is not None:
                # the compared blocks must see the GPU's delay line: P partitions of history + the window's 1024 samples
                nb = KB + (-(-len(ir) // B)) + kN_BLOCKS(B)
            nb = min(nb, i_last * KB + KB)   # no more than the run has processed
            if world == 1:
                all_pos, all_n = pos, n_pos
            else:
                # the other ranks' sources too, but only the blocks the oracle replays (the trajectory is periodic in
                # the block index: absolute indices work as they are)
                first = i_last * KB + KB - nb
                all_pos = wl.trajectories(jf, all_ids, nb, moving=not args.stationary, first_block=first,
                                          ele_override=5 if narrow else None, move_every=args.move_every)
                all_n = None
            base, ok, check = cpu_baseline_and_check(jf, wl, hrir, all_ids, all_pos, all_n, i_last * KB, KB, nb,
                                                     last_mix, groups, G, order,
                                                     reverb=(ir, RV_GAIN) if ir is not None else None)
            out["cpu_baseline"] = base
            out["cpu_baseline"]["gpu_over_cpu"] = value / base["value"]
            out["verified"] = ok
            out["verification"] = check

    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 256 warm-up steps (80 ms) because the first ~100 steps after an idle GPU run 10-15 % slower
    # (clock ramp; profiles/r01_experiments.md), then 1024 timed steps = 131 072 blocks x 1024 sources (0.29 s)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--stationary", action="store_true", help="sources do not move (no crossfade)")
    ap.add_argument("--move-every", type=int, default=1,
                    help="the sources' azimuth advances every n-th block (default 1 = configs[2]'s moving sources; 172 = the "
                         "dwell of the reference's benchmarkTesting): variants, not the default bench line")
    ap.add_argument("--reverb", action="store_true",
                    help="BASELINE.json configs[4]: 256 sources, 128-sample blocks, 2 s convolution-reverb IR "
                         "(partitioned FDL convolution ahead of the spatialiser); not the default bench line")
    ap.add_argument("--realtime", action="store_true",
                    help="with --reverb: one block per call (the audio callback's shape), where the delay line "
                         "is read from HBM: roofline of the per-block multiply-accumulate kernel")
    ap.add_argument("--rv-sources", type=int, default=256,
                    help="with --reverb: sources (512 puts the real-time form's 362 MB delay line past the 256 MB "
                         "Infinity Cache: the HBM-bound measurement)")
    ap.add_argument("--rv-ir-seconds", type=float, default=2.0,
                    help="with --reverb: length of the impulse response (4.0 = media/s1_r1_b_441_mono.wav's: 1379 partitions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="the default line without its \"also\" record (config 5 in batch form and in real time, the stationary variant)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 counter passes")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="HIP events around every n-th fused launch of the timed region (1: around every one)")
    ap.add_argument("--cpu-sample-blocks", type=int, default=256)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # started by a launcher (WORLD_SIZE set): the ranks form a process group and the mix goes through the reduce even
    # when the group has one rank -- the N > 1 code path on a one-GPU box (tests/test_gpu_engine.py)
    use_dist = world_env is not None

    # The secondary configurations of the default line ("also"): config 5 in batch form and in real time, and the stationary
    # variant of the headline -- run after the headline's timed region and verification, in this process, on fresh engines.
    default_line = not (args.reverb or args.stationary or args.realtime or args.move_every != 1 or args.pmc_child
                        or os.environ.get("JF_BENCH_NARROW") is not None)
    also_legs = also_configurations(args) if (default_line and world == 1 and not args.no_also) else {}
    # every leg's hardware counters FIRST: the profiled children must have come and gone before this process touches the GPU
    pmc, pmc_note = precollect_pmc(args, world)
    also_pmc = {name: precollect_pmc(a, world) for name, a in also_legs.items() if name != "reverb_realtime_us"}

    backend = os.environ.get("JF_DIST_BACKEND", "nccl")
    torch = dist = None
    if not args.pmc_child:
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        # JF_DIST_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the multi-rank path
        # on a 1-GPU box (RCCL refuses duplicate devices); the measured configuration is always nccl.
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)

    from jf_load import jf
    # The rank's host thread on the NUMA node its GPU hangs off (include/jefferson.h: jf_pin_thread_to_device; what
    # `numactl --cpunodebind` does for a launcher that does not): launches, the poll of the per-block legs and the pinned
    # buffers stay on one socket.  JF_NO_PIN=1 leaves the thread where the system put it; the line says which.
    host_info = {"numa_node_of_gpu": None, "thread_pinned_to_it": False}
    try:
        host_info["numa_node_of_gpu"] = jf.device_numa_node(local_rank)
        if not os.environ.get("JF_NO_PIN"):
            host_info["thread_pinned_to_it"] = bool(jf.pin_thread_to_device(local_rank))
    except jf.JfError:
        pass
    wl = load_workload()
    gold = os.path.join(ROOT, "tests", "golden", "kemar_hrir_710x2x128_i16.npy")
    hrir = np.load(gold).astype(np.float32) / np.float32(32768.0)

    ctx = {"torch": torch, "dist": dist, "jf": jf, "wl": wl, "hrir": hrir, "rank": rank, "local_rank": local_rank, "world": world,
           "use_dist": use_dist, "backend": backend, "host_info": host_info}
    t_head = time.perf_counter()
    out = run_leg(args, ctx, pmc, pmc_note)
    if args.pmc_child:
        return
    if rank == 0 and out is not None:
        if also_legs:
            also = {"what": "the other single-GPU configurations, run after the headline's timed region and verification in the same "
                            "process on fresh engines; the headline's value / config / roofline above are untouched by them",
                    "headline_leg_s": time.perf_counter() - t_head}
            for name, a in also_legs.items():
                t_leg = time.perf_counter()
                try:
                    if name == "reverb_realtime_us":
                        rv = also_legs["reverb"]
                        n_ir = int(round(rv.rv_ir_seconds * 44100))
                        also[name] = realtime_reverb_call_us(jf, hrir, rv.rv_sources, 128, reverb_response(n_ir), RV_GAIN)
                    else:
                        full = run_leg(a, ctx, *also_pmc[name])
                        also[name] = brief(full, ("kernel", "bound", "avg_stage_ms", "algorithmic_bytes_per_step", "achieved", "peak",
                                                  "unit", "frac", "traffic", "traffic_is", "frac_of_achievable_6300", "level")
                                           if name == "reverb" else ("kernel", "bound", "avg_launch_ms", "achieved", "peak", "unit", "frac", "frac_pmc", "hbm"))
                except Exception as ex:   # a secondary leg must not cost the headline its line
                    also[name] = {"error": f"{type(ex).__name__}: {ex}"}
                if isinstance(also[name], dict):
                    also[name]["leg_s"] = time.perf_counter() - t_leg
            out["also"] = also
        elif default_line and world > 1:
            out["also"] = None
            out["also_null_because"] = "the secondary configurations are single-GPU ones: reported by the N = 1 line"
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
