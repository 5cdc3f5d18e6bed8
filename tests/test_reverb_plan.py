"""The schedule of the non-uniformly partitioned reverb (jf_host.cpp: host_reverb_schedule, through jf_debug_reverb_schedule;
no GPU): runs of calls of random sizes are replayed against a MODEL of the engine's rings in the order the kernels run
(jf_reverb.hip: launch_stage), and every read is checked to find what it must find:

* every block's wet signal is formed exactly once -- by FULL of its big block, or by the head + TAIL of its big block;
* a head (2 M partitions of one block: two big partitions' worth of taps) reads the small spectra of the 2 M - 1 blocks before
  it: they were transformed (this call or an earlier one) and their slots in the ring of 2 M + maxK slots have not been
  written since;
* X_m is formed exactly once, from 2 M blocks of input that are in the dry ring (earlier calls) or in the running call;
* a product reads spectra X_{m-q} that exist (or lie before the start) and whose ring slots still hold THEM;
* a block reads TAIL of ITS big block from the fut ring (four places): the collision this test was written after;
* one-block calls may hand X_m and TAIL(m + 1) to a side stream when they complete big block m - 1 (jf_engine_reverb.cpp:
  run_reverb_stage): replayed as "formed at once" -- the engine's stream waits for that stream before the next call that is
  not such a one-block call and before the next hand-over, so nothing of it is read earlier than that.
"""
import numpy as np
import pytest


def _schedule(jf, j0, K, M, fut_m):
    return jf.reverb_schedule(j0, K, M, fut_m)


def _replay(jf, sizes, M, P1, max_k, side=False):
    steps_max = max_k // M + 1
    R1, Rn, Fn, Rg = P1 + 16 + steps_max + 4, steps_max + 3, 4, 2 * M + max_k
    x_done = {}            # m -> call index that formed X_m
    x_slot = {}            # ring slot -> m it holds
    dry_blk = {}           # ring position (block granularity, Rn * M places) -> absolute block whose samples lie there
    small = {}             # head ring slot -> absolute block whose small spectrum lies there
    fut = {}               # fut place -> m whose TAIL lies there
    wet_by = {}            # absolute block -> how its wet signal was formed
    j0, fut_m, head = 0, 1, 0          # TAIL(0), TAIL(1): sums over the time before the start (zeros)
    for call, K in enumerate(sizes):
        assert 1 <= K <= max_k
        s = _schedule(jf, j0, K, M, fut_m)
        assert s["n_tr"] <= steps_max and s["n_mid"] <= steps_max
        j1 = j0 + K

        def need_x(m, why):
            if m <= 0:
                return          # before the start: zeros
            assert m in x_done, (why, m, j0, K)
            assert x_slot.get(m % R1) == m, ("X slot overwritten", why, m, x_slot.get(m % R1))

        def tail(m):                # TAIL(m) = sum_{q = 2 .. P1} X_{m+1-q} H'_q
            for q in range(2, P1 + 1):
                need_x(m + 1 - q, ("TAIL", m))
            fut[m % Fn] = m

        def small_path(kb, kn):
            for k in range(kb, kb + kn):
                j = j0 + k
                for p in range(2 * M):                # the head's partitions: spectra of blocks j - p
                    jj = j - p
                    if jj < 0:
                        continue
                    assert small.get((head + k - p) % Rg) == jj, ("small spectrum missing", j, jj, small.get((head + k - p) % Rg))
                m = j // M
                # (big blocks 0 and 1 have no TAIL: their places must still hold the zeros of the reset)
                assert fut.get(m % Fn) == (m if m > 1 else None), ("fut place holds another big block's TAIL", j, m, fut.get(m % Fn))
                assert j not in wet_by
                wet_by[j] = "head+tail"

        # ---- the kernels' order (launch_stage)
        if s["tail_early"] >= 0:
            assert s["tail_early"] == j0 // M and s["tail_early"] > 1
            tail(s["tail_early"])
        # transform kernel: every block but the skipped ones writes its samples to the dry ring; the copy-only ones nothing else
        for k in range(K):
            if s["skip_lo"] <= k < s["skip_hi"]:
                continue
            j = j0 + k
            dry_blk[j % (Rn * M)] = j
            if not (s["copy_lo"] <= k < s["copy_hi"]):
                small[(head + k) % Rg] = j
        split = s["n_ranges"] > 1
        if split:
            small_path(s["kb0"], s["kn0"])
        hand_over = side and K == 1 and s["n_tr"] > 0      # the transform (and a TAIL) go to the side stream
        for i in range(s["n_tr"]):
            m = s["m_lo"] + i
            assert j0 < M * m <= j1 and m not in x_done
            for j in range(M * (m - 2), M * m):       # its 2 M blocks of input
                if j < 0:
                    continue
                if j < j0 or hand_over:
                    assert dry_blk.get(j % (Rn * M)) == j, ("dry ring lost a block", m, j, dry_blk.get(j % (Rn * M)))
                # (blocks of the running call are read from the signal itself -- on the side stream from the dry ring too)
            x_done[m] = call
            x_slot[m % R1] = m
        for i in range(s["n_mid"]):
            m = s["ma"] + i
            assert M * m >= j0 and M * (m + 1) <= j1
            for q in range(P1 + 1):
                need_x(m + 1 - q, ("FULL", m))
            for j in range(M * m, M * (m + 1)):
                assert j not in wet_by
                wet_by[j] = "full"
        if s["tail_late"] >= 0:
            tail(s["tail_late"])
        new_fut_m = s["fut_m"]
        if hand_over:
            assert s["tail_late"] < 0 and s["n_mid"] == 0
            mb = j0 // M
            assert s["m_lo"] == mb + 1 and j1 == M * (mb + 1)
            assert fut.get((mb + 2) % Fn) not in (mb, mb + 1), "the side stream's TAIL lands on a place still being read"
            if fut_m < mb + 1:      # the run of one-block calls began inside this big block: TAIL(mb + 1) as well
                tail(mb + 1)        # (the next call waits for the side stream)
            tail(mb + 2)
            new_fut_m = max(new_fut_m, mb + 2)
        if split:
            small_path(s["kb1"], s["kn1"])
        else:
            small_path(s["kb0"], s["kn0"])
        # the ranges and the whole big blocks tile the call
        covered = sorted(j for j in wet_by if j0 <= j < j1)
        assert covered == list(range(j0, j1))
        j0, fut_m, head = j1, new_fut_m, (head + K) % Rg
    return wet_by


@pytest.mark.parametrize("M,P1,max_k", [(16, 43, 256), (16, 3, 31), (8, 6, 64), (16, 1, 16), (8, 2, 1)])
def test_runs_of_calls_find_what_they_read(jf, M, P1, max_k):
    rng = np.random.default_rng(M * 1000 + P1 + max_k)
    shapes = [[1] * (6 * M + 3), [max_k] * 12, [6, max_k if max_k > 6 else 1] * 4, [M] * 9 if M <= max_k else [1],
              [M - 1, 1, M + 1 if M + 1 <= max_k else 1, 2 * M if 2 * M <= max_k else 1] * 5]
    for _ in range(40):
        shapes.append(rng.integers(1, max_k + 1, size=40).tolist())
    for sizes in shapes:
        for side in (False, True):
            wet = _replay(jf, [min(int(k), max_k) for k in sizes], M, P1, max_k, side=side)
            if max_k >= 2 * M:
                assert "full" in wet.values() or max(sizes) < M
    # one-block calls among the others: runs of ones long enough to cross big-block boundaries
    for _ in range(30):
        sizes = []
        while len(sizes) < 60:
            sizes += [1] * int(rng.integers(1, 3 * M)) + rng.integers(1, max_k + 1, size=int(rng.integers(1, 4))).tolist()
        _replay(jf, [min(int(k), max_k) for k in sizes], M, P1, max_k, side=True)


def test_the_cases_the_gpu_tests_run(jf):
    """(6, 64) at M = 16: the blocks in front of the first whole big block (big block 0) and TAIL of big block 4 share a place of
    the fut ring -- the range in front must be finished first (the replay above fails if the order is the other way round)."""
    s = jf.reverb_schedule(6, 64, 16, 1)
    assert (s["kn0"], s["n_mid"], s["kb1"], s["kn1"], s["tail_late"], s["tail_early"]) == (10, 3, 58, 6, 4, -1)
    assert (s["copy_lo"], s["copy_hi"], s["skip_lo"], s["skip_hi"]) == (10, 27, 10, 27)   # the last 2 M - 1 = 31 transformed
    # aligned calls of whole big blocks: nothing goes through the head, no TAIL at all
    s = jf.reverb_schedule(512, 256, 16, 31)
    assert (s["n_ranges"], s["kn0"], s["kn1"], s["n_mid"], s["n_tr"], s["tail_early"], s["tail_late"]) == (2, 0, 0, 16, 16, -1, -1)
    # per-block calls, everything in line: X_m behind the block that completes a big block, TAIL in front of the next one
    # (with the side stream TAIL(2) is there by then: fut_m = 2, nothing in front)
    assert jf.reverb_schedule(31, 1, 16, 1)["n_tr"] == 1 and jf.reverb_schedule(31, 1, 16, 1)["tail_early"] == -1
    assert jf.reverb_schedule(32, 1, 16, 1)["tail_early"] == 2 and jf.reverb_schedule(32, 1, 16, 1)["n_tr"] == 0
    assert jf.reverb_schedule(32, 1, 16, 2)["tail_early"] == -1
