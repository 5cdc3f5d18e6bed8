"""world_size-2 gloo test of the multi-GPU decomposition: sources are sharded
over ranks with no data-path collective, the per-rank stereo mixes are summed with one
reduce to rank 0.  On CPU the per-rank compute is the oracle (tests may use it as the
stand-in); what is under test is the sharding + reduce logic bench.py uses on RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load_workload():
    import importlib.util
    spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "jefferson-2.0_amd", "workload.py"))
    wl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wl)
    return wl


def _worker(rank, world, port, n_total, K, B, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from jf_load import jf
    wl = _load_workload()
    hrir = np.load(os.path.join(ROOT, "tests", "golden", "kemar_hrir_710x2x128_i16.npy")).astype(np.float32) / 32768
    lo, hi = wl.shard_range(n_total, world, rank)
    ids = np.arange(lo, hi)
    eng = oracle_lib.Engine(B, 512, len(ids), hrir)
    for s, sid in enumerate(ids):
        eng.set_signal(s, wl.source_signal_and_start(sid, 3000)[0])
    pos = wl.trajectories(jf, ids, K)
    mix = torch.from_numpy(eng.process_batch(pos, n_threads=1))
    dist.reduce(mix, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, mix.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_mix_equals_single_process(tmp_path, hrir, jf):
    import oracle_lib
    wl = _load_workload()
    n_total, K, B, world = 5, 4, 128, 2   # ragged split: 3 + 2 sources
    out_path = str(tmp_path / "mix.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_total, K, B, out_path), nprocs=world, join=True)
    got = np.load(out_path)

    ids = np.arange(n_total)
    eng = oracle_lib.Engine(B, 512, n_total, hrir)
    for s in ids:
        eng.set_signal(int(s), wl.source_signal_and_start(s, 3000)[0])
    mix, part = eng.process_batch(wl.trajectories(jf, ids, K), want_partial=True, n_threads=1)
    # same association as the sharded job: (sum of rank 0's sources) + (sum of rank 1's)
    lo0, hi0 = wl.shard_range(n_total, world, 0)
    expect = part[lo0:hi0].sum(axis=0, dtype=np.float32) * 0
    for r in range(world):
        lo, hi = wl.shard_range(n_total, world, r)
        acc = np.zeros_like(mix)
        for s in range(lo, hi):
            acc = acc + part[s]
        expect = expect + acc
    assert np.array_equal(got, expect)
    assert np.abs(got - mix).max() < 1e-6  # vs the strictly serial single-process order
