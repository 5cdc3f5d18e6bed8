"""The one-launch real-time kernel (per-block calls with few sources) against the batch pipeline and
the oracle, on a real MI355X."""
import numpy as np
import pytest

import oracle_lib
from conftest import assert_within, sum_tol

pytestmark = pytest.mark.gpu

TOL32 = 4e-7


def _setup(jf, hrir, castanets, S, B, rt_max):
    e = jf.Engine(B, 512, S, hrir=hrir)
    e.set_rt_max_sources(rt_max)
    for s in range(S):
        e.set_signal(s, 0.3 * np.roll(castanets, 777 * s)[: 9000 + 101 * s])
    return e


def _run(e, S, K, oracle=None):
    out, ref = [], []
    for b in range(K):
        for s in range(S):
            ele, azi, r = -35 + (6 * s) % 120, (11 * s + 4 * (b // 2) * (1 + s % 3)) % 360, 0.4 + 0.2 * (s % 5)
            e.set_spherical(s, ele, azi, r)
            if oracle is not None:
                oracle.set_spherical(s, ele, azi, r)
        out.append(e.process_block())
        if oracle is not None:
            ref.append(oracle.process_block())
    return np.array(out), (np.array(ref) if oracle is not None else None)


@pytest.mark.parametrize("B", [128, 256])
@pytest.mark.parametrize("S", [1, 7, 8, 16])
def test_realtime_kernel_equals_batch_pipeline(jf, hrir, castanets, S, B):
    """Same items, same arithmetic, same sum order for S <= 8 (one source per wave, eight waves to the workgroup, waves
    added in order = the mix kernel's order): bit-identical to prep + fused + mix with one block per call.  16 sources
    are two workgroups whose blocks the host adds: another association of the same sum, a few ulps apart."""
    rt = _setup(jf, hrir, castanets, S, B, 16)
    ref = _setup(jf, hrir, castanets, S, B, 0)
    a, _ = _run(rt, S, 9)
    b, _ = _run(ref, S, 9)
    assert rt.last_kernels()[-1] == "rt_block_kernel<%d,8>" % (B // 64)
    rt.close()
    ref.close()
    assert np.abs(a).max() > 0.01
    if S <= 8:
        assert np.array_equal(a, b)
    else:
        assert np.abs(a - b).max() <= TOL32


def test_realtime_kernel_more_sources_than_waves(jf, hrir, castanets):
    """37 sources: five workgroups of eight waves, the last with five sources -- a different association of the
    same sum, so compared with the oracle to tolerance."""
    S, B, K = 37, 256, 6
    e = _setup(jf, hrir, castanets, S, B, 40)
    o = oracle_lib.Engine(B, 512, S, hrir)
    for s in range(S):
        o.set_signal(s, 0.3 * np.roll(castanets, 777 * s)[: 9000 + 101 * s])
    got, want = _run(e, S, K, o)
    e.close()
    assert np.abs(got - want).max() <= TOL32 * 4


@pytest.mark.parametrize("S,B", [(100, 256), (256, 128), (300, 64), (1030, 256), (2100, 128)])
def test_realtime_kernel_many_workgroups(jf, hrir, castanets, S, B):
    """Default settings: up to 8192 sources go through the one-launch kernel with one workgroup per 8 sources up to 512
    sources and per 16 beyond (at most 128 workgroups: with 2100 sources a wave takes two), the workgroups' blocks added
    on the host in order.  Against the oracle and against the batch pipeline (rt_max = 0): same items, other association
    of the sum."""
    K = 5
    a = _setup(jf, hrir, castanets, S, B, 8192)
    b = _setup(jf, hrir, castanets, S, B, 0)
    o = oracle_lib.Engine(B, 512, S, hrir)
    for s in range(S):
        o.set_signal(s, 0.3 * np.roll(castanets, 777 * s)[: 9000 + 101 * s])
    got, want = _run(a, S, K, o)
    ref, _ = _run(b, S, K)
    assert a.last_kernels()[-1] == "rt_block_kernel<%d,%d>" % (B // 64, 8 if S <= 512 else 16)
    assert b.last_kernels()[-1].startswith("mix")
    a.close()
    b.close()
    assert np.abs(want).max() > 0.1
    assert_within(got, want, sum_tol(TOL32, S), f'real-time kernel S={S} B={B}: vs oracle32')
    assert_within(got, ref, sum_tol(TOL32, S), f'real-time kernel S={S} B={B}: vs the batch pipeline')


def test_realtime_kernel_state_carries_into_batch_calls(jf, hrir, castanets):
    """Windows, play positions and crossfade state written by the real-time kernel are the ones the
    batch kernels continue from (and vice versa)."""
    S, B = 3, 256
    a = _setup(jf, hrir, castanets, S, B, 16)
    b = _setup(jf, hrir, castanets, S, B, 0)
    pos = np.zeros((8, S, 5), np.float32)
    for k in range(8):
        for s in range(S):
            pos[k, s] = jf.position_from_spherical(10 * s, (20 * s + 5 * k) % 360, 0.6)
    outs = []
    for e in (a, b):
        o = []
        for k in range(3):                       # per-block calls
            for s in range(S):
                e.set_spherical(s, pos[k, s, 0], pos[k, s, 1], 0.6)
            o.append(e.process_block())
        e2 = e.process_batch(pos[3:4])           # a batch call of one block (maxK = 1)
        o.append(e2[0])
        for k in range(4, 8):                    # and back
            for s in range(S):
                e.set_spherical(s, pos[k, s, 0], pos[k, s, 1], 0.6)
            o.append(e.process_block())
        outs.append(np.array(o))
        e.close()
    assert np.array_equal(outs[0], outs[1])


def test_setters_from_another_thread(jf, hrir, castanets):
    """graphics.cu:378 writes the position from the GLUT thread while the PortAudio thread runs the
    callback (no lock in the reference).  Here the setters are mutex-protected: hammer them from a
    second thread during processing; every block must be a valid rendering of SOME latched position
    (finite, bounded), and the last block must match the final position exactly once the writer stops."""
    import threading
    e = _setup(jf, hrir, castanets, 2, 256, 16)
    stop = threading.Event()

    def writer():
        k = 0
        while not stop.is_set():
            e.set_cartesian(0, 0.5 * np.cos(0.01 * k), 0.1, 0.5 * np.sin(0.01 * k))
            e.set_spherical(1, 10, k % 360, 0.8)
            k += 1

    t = threading.Thread(target=writer)
    t.start()
    try:
        for _ in range(300):
            y = e.process_block()
            assert np.isfinite(y).all() and np.abs(y).max() < 2.0
    finally:
        stop.set()
        t.join()
    pa, pb = e.get_position(0), e.get_position(1)
    e.process_block()                 # latches the final positions (crossfade block)
    y = e.process_block()             # stationary block at the final positions
    ref = _setup(jf, hrir, castanets, 2, 256, 16)
    # replay: same number of blocks consumed, then the same final positions
    for _ in range(300):
        ref.process_block()
    ref.set_cartesian(0, float(pa[3]), float(pa[4]), float(pa[5]))
    ref.set_spherical(1, float(pb[0]), float(pb[1]), float(pb[2]))
    ref.process_block()
    assert np.array_equal(ref.process_block(), y)
    e.close()
    ref.close()


def test_completion_words_under_load_every_block_checked(jf, hrir, castanets):
    """Per-block calls learn that a block has landed from words the kernel stores into host memory behind the
    block (no stream synchronisation).  A word that overtook a workgroup's stores would hand out the previous
    block's frames: 1500 consecutive blocks of 300 sources (several workgroups, four storing waves each), with a
    second engine's batch launches keeping the GPU busy meanwhile, every block compared with an engine that takes
    the three-launch pipeline (a stream synchronisation per block)."""
    S, B, K = 300, 128, 1500
    rt = _setup(jf, hrir, castanets, S, B, 8192)
    ref = _setup(jf, hrir, castanets, S, B, 0)
    for e in (rt, ref):
        for s in range(S):
            e.set_spherical(s, -35 + (6 * s) % 120, (11 * s) % 360, 0.4 + 0.2 * (s % 5))
    Sb, Kb = 512, 64
    bg = jf.Engine(256, 512, Sb, hrir=hrir, max_batch_blocks=Kb)
    for s in range(Sb):
        bg.set_signal(s, 0.3 * np.roll(castanets, 311 * s)[:20000])
    ele = np.broadcast_to(10.0 * (np.arange(Sb) % 9), (Kb, Sb)).astype(np.float32)
    azi = ((7.0 * np.arange(Sb)[None, :] + np.arange(Kb)[:, None]) % 360).astype(np.float32)
    bg.upload_positions(jf.positions_from_spherical(ele, azi, np.ones((Kb, Sb), np.float32)))
    worst, loud = 0.0, 0.0
    for b in range(K):
        if b % 4 == 0:
            bg.batch_run(0, Kb)  # asynchronous: queued on the other engine's stream
        got = rt.process_block()
        want = ref.process_block()
        worst = max(worst, float(np.abs(got - want).max()))
        loud = max(loud, float(np.abs(want).max()))
    bg.synchronize()
    assert any("rt_block_kernel" in k for k in rt.last_kernels())
    assert not any("rt_block_kernel" in k for k in ref.last_kernels())
    bg.close()
    rt.close()
    ref.close()
    assert loud > 0.05
    assert worst <= sum_tol(TOL32, 64) * max(1.0, loud)


def test_the_calling_thread_can_be_put_on_the_gpus_numa_node():
    """jf_device_numa_node / jf_pin_thread_to_device (include/jefferson.h): the node the system reports for the device (or -1)
    and, where it reports one, an affinity mask inside that node's CPUs afterwards -- in a child process, so that the test
    runner's own affinity stays what it was.  A device that does not exist is an error, not a guess."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ('import os, sys\nsys.path.insert(0, %r)\nfrom jf_load import jf\n'
            'n = jf.device_numa_node(0)\nbefore = os.sched_getaffinity(0)\nok = jf.pin_thread_to_device(0)\n'
            'after = os.sched_getaffinity(0)\nprint(n, int(ok), len(before), len(after), int(after <= before))\n'
            'if n >= 0 and ok:\n'
            '    lst = open("/sys/devices/system/node/node%%d/cpulist" %% n).read().strip()\n'
            '    cpus = set()\n'
            '    for part in lst.split(","):\n'
            '        a, _, b = part.partition("-")\n'
            '        cpus |= set(range(int(a), int(b or a) + 1))\n'
            '    print(int(after <= cpus))\n'
            'try:\n    jf.device_numa_node(99)\n    print("NOERR")\nexcept jf.JfError as ex:\n    print("ERR", ex.code)\n' % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    lines = r.stdout.decode().split("\n")
    n, ok, n_before, n_after, inside = (int(x) for x in lines[0].split())
    assert n >= -1 and inside == 1 and 1 <= n_after <= n_before
    if n >= 0 and ok:
        assert lines[1].strip() == "1"          # only CPUs of the device's node are left
        assert lines[2].startswith("ERR")
    else:
        assert n_after == n_before              # nothing was changed
        assert lines[1].startswith("ERR")


def test_per_block_calls_continue_from_where_a_batch_call_left_the_sources(jf, hrir, castanets):
    """jf_process_batch is n callbacks in one call: afterwards the sources stand where the last of them read them (as if the
    setters had been called before each block), so a per-block call that follows WITHOUT a setter call continues from there --
    no crossfade back to what the setters held before the batch.  Found by tests/test_gpu_random_sessions.py: the engine used
    to keep the pre-batch positions.  jf_sources_set_latched is the same thing said explicitly; jf_batch_run leaves the sources
    alone."""
    S, B, K = 3, 256, 5
    e = jf.Engine(B, 512, S, hrir=hrir, max_batch_blocks=K)
    o = oracle_lib.Engine(B, 512, S, hrir)
    for s in range(S):
        sig = 0.3 * castanets[2000 * s: 2000 * s + 9000]
        e.set_signal(s, sig)
        o.set_signal(s, sig)
        e.set_spherical(s, 0, 10 * s, 1.0)
        o.set_spherical(s, 0, 10 * s, 1.0)
    pos = np.zeros((K, S, 5), np.float32)
    for k in range(K):
        for s in range(S):
            pos[k, s] = jf.position_from_spherical(20 + 5 * s, 100 + 30 * s + 7 * k, 0.8)
    assert_within(e.process_batch(pos), o.process_batch(pos), sum_tol(TOL32, S), 'latched: batch')
    for s in range(S):
        assert np.array_equal(e.get_position(s)[[0, 1, 3, 4, 5]], pos[K - 1, s])      # {ele, azi, r, x, y, z}
    a, b = e.process_block(), o.process_block()          # no setter call in between
    assert np.abs(b).max() > 0.002
    assert_within(a, b, sum_tol(TOL32, S), 'latched: block after batch')
    # the same through the explicit entry point, after a run of an uploaded trajectory
    e.upload_positions(pos)
    e.batch_run(0, K)
    e.synchronize()
    want = o.process_batch(pos)
    assert_within(e.read_device(e.mix_device_ptr(), (K, 2 * B)), want, sum_tol(TOL32, S), 'latched: run')
    e.set_latched(pos[2])                                 # "the sources stand at block 2's positions"
    for s in range(S):
        o.set_spherical(s, 20 + 5 * s, 100 + 30 * s + 7 * 2, 0.8)
    a, b = e.process_block(), o.process_block()
    e.close()
    o.close()
    assert_within(a, b, sum_tol(TOL32, S), 'latched: explicit')


def test_a_reset_or_a_new_signal_right_before_a_block(jf, hrir, castanets):
    """jf_source_reset / jf_source_set_signal and, at once, the next block -- 600 times, through the callback (a block in
    flight while the state is rewritten) and through jf_process_block.  The engine's stream is a non-blocking stream: its
    memsets and small copies must go through THAT stream, or the block's kernel may run before them (it did, one session in
    twelve, when they went through the null stream: found by tests/test_gpu_random_sessions.py run over many seeds)."""
    S, B = 5, 256
    rng = np.random.default_rng(5)
    for use_callback in (True, False):
        e = jf.Engine(B, 512, S, hrir=hrir)
        o = oracle_lib.Engine(B, 512, S, hrir)
        for s in range(S):
            sig = 0.4 * castanets[1500 * s:1500 * s + 5000]
            e.set_signal(s, sig)
            o.set_signal(s, sig)
            e.set_spherical(s, 10 * s, 60 * s, 0.7)
            o.set_spherical(s, 10 * s, 60 * s, 0.7)
        prev = np.zeros(2 * B, np.float32)
        worst = 0.0
        for k in range(300):
            s = int(rng.integers(0, S))
            if k % 3 == 0:
                e.reset(s)
                o.reset(s)
            elif k % 3 == 1:
                a = int(rng.integers(0, 20000))
                sig = 0.4 * castanets[a:a + int(rng.integers(50, 4000))]
                e.set_signal(s, sig)
                o.set_signal(s, sig)
            if use_callback:
                got = e.callback()
                worst = max(worst, float(np.abs(got - prev).max()))
                prev = o.process_block()
            else:
                worst = max(worst, float(np.abs(e.process_block() - o.process_block()).max()))
        if use_callback:
            rc, last = e.collect_block()
            assert rc == 0
            worst = max(worst, float(np.abs(last - prev).max()))
        e.close()
        o.close()
        assert worst <= sum_tol(TOL32, S), (use_callback, worst)
