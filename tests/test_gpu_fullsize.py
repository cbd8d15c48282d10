"""GPU parity at the sizes BASELINE.json's configs state: the HIP path through the C ABI
against the CPU oracle, bit-exact on EVERY output of every op of the stream, plus whole-state digests
(fpx_state_digest vs the oracle's fpo_state_digest: vote rounds, vote values, ballots, acceptor scalars,
proxy-leader tallies -- equal digests <=> equal state) and sampled direct readbacks.

Run on the MI355X box: python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()
    return frankenpaxos_amd


def run_both(fa, oracle, script, sample_slots, **kw):
    gpu = fa.Context(fa.make_config(**kw))
    ref = oracle.System(oracle.make_config(**kw))
    out_g = W.run_script(gpu, script)
    out_r = W.run_script(ref, script)
    W.assert_same_outputs(out_g, out_r)
    dg, dr = gpu.state_digest(), ref.state_digest()
    np.testing.assert_array_equal(dg, dr, err_msg="state digests (vote_round, vote_value, ballot, promised, "
                                                  "max_voted, tallies, log)")
    pg, mg = gpu.read_scalars()
    pr, mr = ref.read_scalars()
    np.testing.assert_array_equal(pg, pr)
    np.testing.assert_array_equal(mg, mr)
    for s in sample_slots:
        assert gpu.read_tally(int(s)) == ref.read_tally(int(s)), "tally of slot %d" % s
    for g, r in ((0, 0), (gpu.ngroups - 1, gpu.R - 1)):
        a, b = gpu.read_acceptor(g, r), ref.read_acceptor(g, r)
        assert a[:2] == b[:2]
        for x, y in zip(a[2:], b[2:]):
            np.testing.assert_array_equal(x, y)
    chosen = sum(int(o[2].sum()) for o in out_g if o[0] == "fused") + \
        sum(int(o[8].sum()) for o in out_g if o[0] == "k1k2")
    gpu.close()
    return out_g, chosen


# ---------------------------------------------------------------------------------------------------
# the headline grid: 2^20 slots x 256 acceptors, adversarial stream of SURVEY.md 8(d), seeds 1-3
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_headline_adversarial_1m_slots_256_replicas(fa, oracle, seed, ballot_mode):
    S, R = 1 << 20, 256
    script = W.adversarial_script(S, R, 128, seed, epochs=64, fused=True, subsets=W.fast_subsets)
    out, chosen = run_both(fa, oracle, script, range(0, S, 65537), num_slots=S, num_replicas=R, f=127,
                           ballot_mode=ballot_mode, tally_ways=8)
    nacks = sum(int((o[5] >= 0).sum()) for o in out if o[0] == "fused")
    assert chosen >= S and nacks > 0          # every slot got chosen at least once; stale rounds were Nacked


@pytest.mark.parametrize("ballot_mode,R", [(0, 256), (1, 256), (0, 255), (1, 255)])
def test_headline_size_thrifty_runs(fa, oracle, ballot_mode, R):
    """the adversarial stream at 2^20 slots with every Phase2a sent to a RUN of neighbouring acceptors (rotating
    sector-aligned windows of 120 .. 128: the packed walk of k_phase2, two rows per wavefront step) -- first proposals
    take the packed walk, the epochs that re-propose voted slots fall back to the row-at-a-time walk on the device's
    own verdict (k_runs_check), pre-promised acceptors Nack inside packed steps"""
    S = 1 << 20
    script = W.adversarial_script(S, R, 128, 5 + ballot_mode, epochs=64, fused=True, subsets=W.run_subsets)
    out, chosen = run_both(fa, oracle, script, range(0, S, 65537), num_slots=S, num_replicas=R, f=127,
                           ballot_mode=ballot_mode, tally_ways=8)
    nacks = sum(int((o[5] >= 0).sum()) for o in out if o[0] == "fused")
    assert chosen > S // 32 and nacks > 0        # (windows of 120 .. 128 positions: about one in nine holds a quorum of 128)


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: compartmentalized MultiPaxos, 2x2 grid quorums, 1M slots x 16 acceptor groups
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode,fused", [(0, True), (1, True), (0, False)])
def test_config3_1m_slots_16_groups_of_2x2_grids(fa, oracle, ballot_mode, fused):
    S = 1 << 20
    script = W.adversarial_script(S, 4, 2, 33 + ballot_mode, epochs=64, fused=fused, ngroups=16,
                                  subsets=W.fast_subsets)
    out, chosen = run_both(fa, oracle, script, range(0, S, 65521), num_slots=S, num_replicas=4, num_groups=16,
                           quorum_kind=2, grid_rows=2, grid_cols=2, ballot_mode=ballot_mode, tally_ways=8)
    # a 2x2 grid write quorum needs one acceptor of each row: some random target subsets miss a row
    assert 0 < chosen


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] at its stated size, unfused pipeline too (the fused one is in test_gpu_parity.py)
# ---------------------------------------------------------------------------------------------------
def test_config2_64k_slots_3_acceptors_adversarial(fa, oracle):
    S = 65536
    for fused in (True, False):
        script = W.adversarial_script(S, 3, 2, 17, epochs=64, fused=fused)
        run_both(fa, oracle, script, range(0, S, 4099), num_slots=S, num_replicas=3, f=1, tally_ways=8)


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: Mencius, 256 rotating leader groups, 4M slots, 3 acceptors per group (f = 1)
# ---------------------------------------------------------------------------------------------------
def mencius_stream(S, L, R, epochs, seed):
    """A Mencius-shaped stream (mencius/Leader.scala:342-345, 455; ProxyLeader.scala:231-234): slot s belongs to
    leader group s % L.  Per epoch (a band of rows of the log) about half of the leader groups have commands and
    propose them in their own slots; the others have nothing to say and skip their slots of the band with noop
    ranges (one or two Phase2aNoopRange per group and epoch).  Leader groups move through rounds independently:
    now and then a competing leader of a group pre-promises some acceptors (its Phase2a's of that epoch get
    Nacked by them), then takes over in the next epoch.  Yields ops for either backend."""
    rng = W.Rng(seed)
    rows_per = (S // L) // epochs
    rounds = np.zeros(L, np.int64)
    pending = {}
    for e in range(epochs):
        nprng = rng.np_rng()
        for lg, rnd in pending.items():            # the challengers of the last epoch finish Phase 1 and take over
            rounds[lg] = rnd
            yield ("phase1a", int(lg), int(rnd), 0, None)
        pending = {}
        for lg in nprng.choice(L, size=6, replace=False):
            rnd = W.next_classic_round(2, (e + 1) % 2, int(rounds[lg]))
            pending[int(lg)] = rnd
            pre = W.bits_from_bool(W.random_subsets(nprng, 1, R, 1, 1))[0]
            yield ("phase1a", int(lg), int(rnd), 0, pre)
        active = nprng.random(L) < 0.5
        rows = np.arange(e * rows_per, (e + 1) * rows_per, dtype=np.int64)
        slot = (rows[:, None] * L + np.nonzero(active)[0][None, :]).reshape(-1).astype(np.int32)   # ascending
        rr = rounds[slot % L].astype(np.int32)
        tgt = W.bits_from_bool(W.random_subsets(nprng, len(slot), R, 2, R))
        yield ("fused", slot, rr, W.steady_values(slot), tgt)
        starts, ends, rnds = [], [], []
        for lg in np.nonzero(~active)[0]:
            lo, hi = e * rows_per, (e + 1) * rows_per
            cut = int(nprng.integers(lo, hi + 1))
            for a, b in ((lo, cut), (cut, hi)):
                if a < b or nprng.random() < 0.1:   # now and then an empty range, too
                    starts.append(a * L + lg)
                    ends.append(min(S, (b - 1) * L + lg + 1) if b > a else a * L + lg)
                    rnds.append(rounds[lg])
        order = nprng.permutation(len(starts))
        i32 = lambda x: np.asarray(x, np.int32)[order]
        tm = W.bits_from_bool(W.random_subsets(nprng, len(starts), R, 2, R)).reshape(len(starts), 1, 4)
        yield ("ranges", i32(starts), i32(ends), i32(rnds), tm)


def test_config5_mencius_256_leader_groups_4m_slots(fa, oracle, row_layout):
    S, L, R = 1 << 22, 256, 3
    kw = dict(num_slots=S, num_replicas=R, num_groups=1, num_leader_groups=L, f=1, tally_ways=4)
    gpu, ref = fa.Context(fa.make_config(**kw)), oracle.System(oracle.make_config(**kw))
    n_ranges = n_chosen_ranges = n_chosen = n_nacks = 0
    for op in mencius_stream(S, L, R, epochs=16, seed=5):
        if op[0] == "phase1a":
            a, b = gpu.acceptor_phase1a(*op[1:]), ref.acceptor_phase1a(*op[1:])
            assert a[0] == b[0] == 0
            np.testing.assert_array_equal(a[1], b[1])
            np.testing.assert_array_equal(a[2], b[2])
        elif op[0] == "fused":
            a, b = gpu.phase2_fused(*op[1:]), ref.phase2_fused(*op[1:])
            assert a[0] == b[0] == 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
            ch = a[1].astype(bool)
            n_chosen += int(ch.sum())
            n_nacks += int((a[4] >= 0).sum())
            # the replicas learn the Chosen commands (Replica.handleChosen + executeLog)
            assert gpu.replica_chosen(op[1][ch], a[3][ch]) == ref.replica_chosen(op[1][ch], a[3][ch])
        else:
            a, b = gpu.noop_ranges_fused(*op[1:]), ref.noop_ranges_fused(*op[1:])
            assert a[0] == b[0] == 0
            for x, y in zip(a[1:], b[1:]):
                np.testing.assert_array_equal(x, y)
            n_ranges += len(op[1])
            n_chosen_ranges += int(a[5].sum())
            for i in np.nonzero(a[5])[0][:48]:      # ChosenNoopRange -> Replica.handleChosenNoopRange (a sample)
                s, e = int(op[1][i]), int(op[2][i])
                assert gpu.replica_chosen_noop_range(s, e) == ref.replica_chosen_noop_range(s, e)
    assert n_ranges > 1024 and 0 < n_chosen_ranges < n_ranges and n_chosen > S // 4 and n_nacks > 0
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
    pg, mg = gpu.read_scalars()
    pr, mr = ref.read_scalars()
    np.testing.assert_array_equal(pg, pr)
    np.testing.assert_array_equal(mg, mr)
    assert gpu.replica_state() == tuple(ref.replica_chosen([], [])[1:])
    for s in range(0, S, 262147):
        assert gpu.read_tally(s) == ref.read_tally(s)
    for lg in (0, 255):
        a, b = gpu.read_acceptor(lg, 2), ref.read_acceptor(lg, 2)
        assert a[:2] == b[:2]
        for x, y in zip(a[2:], b[2:]):
            np.testing.assert_array_equal(x, y)


def _band_on_device(fa, gpu, fused_op, ranges_op, independent):
    """one proxy-leader step through fpx_mencius_band_fused_dev on device-resident arrays; returns the two halves' outputs
    in the order of phase2_fused / noop_ranges_fused"""
    import torch
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    _, slot, rr, val, tgt = fused_op
    _, st_, en_, rn_, tm = ranges_op
    n, k, A = len(slot), len(st_), gpu.cfg.num_groups
    ch, cr, cv, nr = (torch.zeros(n, dtype=torch.uint8, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev),
                      torch.full((n,), -1, dtype=torch.int32, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev))
    vb, nb = (torch.zeros((k, A, 4), dtype=torch.int64, device=dev) for _ in range(2))
    rnr, new, rch = (torch.full((k,), -1, dtype=torch.int32, device=dev), torch.zeros(k, dtype=torch.uint8, device=dev),
                     torch.zeros(k, dtype=torch.uint8, device=dev))
    gpu.mencius_band_fused_dev(d(slot), d(rr), d(val), None if tgt is None else d(tgt.view(np.int64)), ch, cr, cv, nr, d(st_), d(en_),
                               d(rn_), d(np.ascontiguousarray(tm).view(np.int64)), vb, nb, rnr, new, rch, independent=independent)
    st = gpu.sync()
    h = lambda t: t.cpu().numpy()
    return st, (h(ch), h(cr), h(cv), h(nr)), (h(vb).view(np.uint64), h(nb).view(np.uint64), h(rnr), h(new), h(rch))


@pytest.mark.parametrize("dense", [False, True])
def test_config5_band_entry_point_halves_side_by_side(fa, oracle, row_layout, dense):
    """configs[4] at size through fpx_mencius_band_fused_dev -- the commands of the leader groups that
    propose and the noop ranges of those that skip in ONE call (FPX_F_TRUSTED, the leader groups of an epoch's two batches
    are disjoint by construction) -- every output of both halves, the state digest and the acceptors' scalars == the oracle
    running the halves one after the other (mencius/ProxyLeader.scala:216-303).  With target masks on the commands the
    halves run on two streams; dense (every command to every acceptor of its group: no mask) on leader-group-major rows
    the step is TWO launches -- the range chain as the vote kernel's first workgroup, the vote kernel's fold of maxima in
    the fill's grid -- which the context counts."""
    import torch
    S, L, R = 1 << 22, 256, 3
    kw = dict(num_slots=S, num_replicas=R, num_groups=1, num_leader_groups=L, f=1, tally_ways=4)
    gpu, ref = fa.Context(fa.make_config(flags=fa.FPX_F_TRUSTED, **kw)), oracle.System(oracle.make_config(**kw))
    gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    pending, steps = None, 0
    for op in mencius_stream(S, L, R, epochs=16, seed=7):
        if op[0] == "phase1a":
            a, b = gpu.acceptor_phase1a(*op[1:]), ref.acceptor_phase1a(*op[1:])
            assert a[0] == b[0] == 0
            np.testing.assert_array_equal(a[1], b[1])
        elif op[0] == "fused":
            pending = op[:4] + (None,) if dense else op
        else:
            st, cmd, rng_ = _band_on_device(fa, gpu, pending, op, independent=True)
            assert st == 0
            b1, b2 = ref.phase2_fused(*pending[1:]), ref.noop_ranges_fused(*op[1:])
            assert b1[0] == b2[0] == 0
            chosen = b1[1].astype(bool)
            np.testing.assert_array_equal(cmd[0], b1[1])
            np.testing.assert_array_equal(cmd[1][chosen], b1[2][chosen])
            np.testing.assert_array_equal(cmd[2][chosen], b1[3][chosen])
            np.testing.assert_array_equal(cmd[3], b1[4])
            for x, y in zip(rng_, b2[1:]):
                np.testing.assert_array_equal(x, np.asarray(y).reshape(x.shape))
            steps += 1
    assert steps == 16
    import os
    merged = dense and row_layout == "leader-group-major" and not os.environ.get("FPX_BAND_SERIAL")
    assert gpu.band_merged_steps() == (16 if merged else 0)
    np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
    pg, mg = gpu.read_scalars()
    pr, mr = ref.read_scalars()
    np.testing.assert_array_equal(pg, pr)
    np.testing.assert_array_equal(mg, mr)


def test_band_entry_point_when_the_halves_share_a_leader_group(fa, oracle, row_layout, monkeypatch):
    """the halves are NOT independent: leader group 2 proposes commands in some of its slots and skips others with a range
    in the same step, and a range covers a slot that carries a command.  A validating context refuses the caller's
    `independent` (FPX_EORDER, nothing applied: the state digest does not move) and runs the step in order without it;
    a trusted context told the truth (independent = 0) runs it in order too: == the oracle, commands first."""
    import torch
    S, L, R = 1 << 14, 8, 3
    kw = dict(num_slots=S, num_replicas=R, num_groups=1, num_leader_groups=L, f=1, tally_ways=4)
    rows = np.arange(0, 512)
    slot = np.sort(np.concatenate([rows * L + 0, rows * L + 1, rows[:128] * L + 2])).astype(np.int32)
    fused = ("fused", slot, np.zeros(len(slot), np.int32), W.steady_values(slot), W.bits_from_bool(np.ones((len(slot), R), bool)))
    starts = np.array([64 * L + 2, 0 * L + 3, 100 * L + 3], np.int32)      # leader group 2 again: overlaps its commands 64 .. 127
    ends = np.array([400 * L + 2 + 1, 100 * L + 3, 511 * L + 3 + 1], np.int32)
    ranges = ("ranges", starts, ends, np.zeros(3, np.int32), W.bits_from_bool(np.ones((3, R), bool)).reshape(3, 1, 4))
    for flags in (0, fa.FPX_F_TRUSTED):
        gpu, ref = fa.Context(fa.make_config(flags=flags, **kw)), oracle.System(oracle.make_config(**kw))
        gpu.set_stream(torch.cuda.current_stream().cuda_stream)
        for lg in range(L):
            assert gpu.acceptor_phase1a(lg, 0)[0] == 0 and ref.acceptor_phase1a(lg, 0)[0] == 0
        # a validating context checks the caller's word; a TRUSTED one only under FPX_DEBUG_CHECKS=1 (the
        # hazard of a false `independent` there is silent state corruption, include/fpx.h)
        if flags != 0:
            monkeypatch.setenv("FPX_DEBUG_CHECKS", "1")
        before = gpu.state_digest()
        st, cmd, rng_ = _band_on_device(fa, gpu, fused, ranges, independent=True)
        assert st == fa.FPX_EORDER and not cmd[0].any() and not rng_[4].any()
        np.testing.assert_array_equal(gpu.state_digest(), before)
        monkeypatch.delenv("FPX_DEBUG_CHECKS", raising=False)
        st, cmd, rng_ = _band_on_device(fa, gpu, fused, ranges, independent=False)
        assert st == 0
        b1, b2 = ref.phase2_fused(*fused[1:]), ref.noop_ranges_fused(*ranges[1:])
        np.testing.assert_array_equal(cmd[0], b1[1])
        np.testing.assert_array_equal(cmd[2][b1[1].astype(bool)], b1[3][b1[1].astype(bool)])
        for x, y in zip(rng_, b2[1:]):
            np.testing.assert_array_equal(x, np.asarray(y).reshape(x.shape))
        np.testing.assert_array_equal(gpu.state_digest(), ref.state_digest())
        for r in range(R):
            a, b = gpu.read_acceptor(2, r), ref.read_acceptor(2, r)
            assert a[:2] == b[:2]
            for x, y in zip(a[2:], b[2:]):
                np.testing.assert_array_equal(x, y)
        gpu.close()


def test_epaxos_full_size_scenario_with_the_command_log(oracle):
    """BASELINE.json configs[3] end to end AT SIZE with the command log kept (K6 / K7 were only
    held at <= 5 000 messages): a tick of 2^20 commands (K5) -> every slow-path command through the Accept phase
    (fpx_epx_accept, K6: ~800 000 instances, quorum f + 1) -> 2^18 PreAccepts once more at random replicas in old and
    new ballots (fpx_epx_handle_preaccept, K7: re-sent replies, Nacks, Commits sent back, fresh processing) -> a second
    tick on top.  Every output array of every step, the conflict indexes and a sample of the command log equal the
    oracle's."""
    from frankenpaxos_amd.epaxos import EPaxos
    from tests.workloads import random_tick

    n, num_keys, m, NI = 5, 1024, 1 << 20, 1 << 19
    gpu, ref = EPaxos(n, num_keys, num_instances=NI), oracle.EPaxos(n, num_keys, num_instances=NI)
    rng = np.random.default_rng(2020)
    nxt = [0] * n

    def same(a, b, what):
        assert a[0] == b[0] == 0, (what, a[0], b[0])
        for j, (x, y) in enumerate(zip(a[1:], b[1:])):
            np.testing.assert_array_equal(np.asarray(x), np.asarray(y), err_msg="%s output %d" % (what, j))

    # 1. the tick
    leader, number, key, is_set, mask, rank = random_tick(rng, n, num_keys, m, nxt, 3000.0, fifo=False)   # reordering
    tr = np.arange(m, dtype=np.int32)                                    # channels: most answers differ -> slow path
    a, b = (e.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr) for e in (gpu, ref))
    same(a, b, "tick 1")
    fast = a[1].astype(bool)
    assert 0 < fast.sum() < m
    # 2. 2^18 PreAccepts of the tick's instances once more, at random replicas: half in their original ballot (re-sent
    #    replies where the replica voted in it, fresh processing where it never saw the instance, Commit sent back for
    #    the fast-path commits), half in a higher ballot of another replica (processed afresh; Nacked by replicas that
    #    have meanwhile seen a still higher one)
    pick = rng.choice(m, size=1 << 18, replace=False)
    hl, hx = leader[pick], number[pick]
    tgt = rng.integers(1, 1 << n, len(pick)).astype(np.uint8)
    din = rng.integers(0, max(nxt), (len(pick), n)).astype(np.int32)
    din[np.arange(len(pick)), hl] = np.minimum(din[np.arange(len(pick)), hl], hx)     # own column: a plain watermark
    dend = np.zeros(len(pick), np.int32)
    higher = rng.random(len(pick)) < 0.5
    counts = np.zeros(4, np.int64)
    for round_ in range(2):   # second round: the same PreAccepts in the ORIGINAL ballots -- below what round one left
        b_ord = np.where(higher & (round_ == 0), rng.integers(1, 3, len(pick)), 0).astype(np.int32)
        b_rep = np.where(higher & (round_ == 0), rng.integers(0, n, len(pick)), hl).astype(np.int32)
        a, b = (e.handle_preaccept(hl, hx, b_ord, b_rep, key[pick], is_set[pick], (tr[pick] + (2 + round_) * m).astype(np.int32),
                                   din, dend, tgt) for e in (gpu, ref))
        same(a, b, "handle_preaccept %d" % round_)
        counts += [int(np.unpackbits(np.asarray(x)).sum()) for x in a[1:5]]           # ok, resend, nack, commit
    assert all(c > 1000 for c in counts), counts
    # 3. Accept phase of the slow-path commands: every leader proposes in Ballot(3, leader) -- above everything step 2
    #    used -- to f = 2 of the 4 other replicas
    slow = np.nonzero(~fast)[0]
    sl, sx = leader[slow], number[slow]
    others = np.array([[r for r in range(n) if r != L] for L in range(n)])
    two = np.argsort(rng.random((len(slow), n - 1)), axis=1)[:, :2]
    tgt = np.zeros(len(slow), np.uint8)
    for j in range(2):
        tgt |= (1 << others[sl, two[:, j]]).astype(np.uint8)
    a, b = (e.accept(sl, sx, np.full(len(slow), 3, np.int32), sl, (tr[slow] + m).astype(np.int32), tgt, key[slow], is_set[slow])
            for e in (gpu, ref))
    same(a, b, "accept")
    assert int(np.asarray(a[5]).sum()) == len(slow)                                   # all commit: proposer + 2 = f + 1
    # 4. a second tick on top of all that
    leader2, number2, key2, is_set2, mask2, rank2 = random_tick(rng, n, num_keys, 1 << 18, nxt, 64.0, fifo=False)
    a, b = (e.preaccept(leader2, number2, key2, is_set2, mask2, rank2, triple_id=np.arange(1 << 18, dtype=np.int32) + 4 * m)
            for e in (gpu, ref))
    same(a, b, "tick 2")
    # state: every conflict index, and the command log where the steps above left the most different entries
    for r in range(n):
        for k in range(0, num_keys, 7):
            for x, y in zip(gpu.read_index(r, k), ref.read_index(r, k)):
                np.testing.assert_array_equal(x, y)
    for r in range(n):
        for j in rng.choice(m, size=120, replace=False):
            L, x = int(leader[j]), int(number[j])
            assert gpu.read_cmdlog(r, L, x) == ref.read_cmdlog(r, L, x)
            c, d = gpu.read_cmdlog_deps(r, L, x), ref.read_cmdlog_deps(r, L, x)
            assert c[0].tolist() == d[0].tolist() and c[1] == d[1]


# ---------------------------------------------------------------------------------------------------
# size-independent properties at the headline size, no oracle involved: what Paxos itself promises about a batch
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_headline_size_properties(fa, ballot_mode):
    """2^20 slots x 256 acceptors, the GPU alone.  (1) a steady batch chooses every slot with its own value; (2) the
    same batch again changes nothing -- no output, no state (idempotence: a known (slot, round) is ignored,
    ProxyLeader.scala:176-184); (3) the batch in a shuffled order, and (4) in two halves with a stale half in between,
    leave the SAME state (a digest of the whole state: votes, ballots, scalars, tallies) and the same per-slot results;
    (5) a higher round proposing the chosen values again chooses them again, a stale round is Nacked by everybody and
    chooses nothing, and no slot ever reports two different chosen values (safety)."""
    S, R = 1 << 20, 256
    kw = dict(num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode, tally_ways=8)
    slot, rnd, val = W.steady_stream(S)
    rnd = rnd + 2                                            # round 2: leaves room for a stale round below it
    a = fa.Context(fa.make_config(**kw))
    assert a.acceptor_phase1a(0, 2)[0] == 0                  # the leader of round 2 ran Phase 1
    st, ch, cr, cv, nr = a.phase2_fused(slot, rnd, val)
    assert st == 0 and ch.all() and (cr == 2).all() and (cv == val).all() and (nr == -1).all()      # (1)
    d1 = a.state_digest()
    st, ch2, cr2, cv2, nr2 = a.phase2_fused(slot, rnd, val)
    assert st == 0 and not ch2.any() and (cr2 == -1).all() and (cv2 == -1).all()                    # (2)
    np.testing.assert_array_equal(a.state_digest(), d1)

    b = fa.Context(fa.make_config(**kw))
    assert b.acceptor_phase1a(0, 2)[0] == 0
    perm = np.random.default_rng(9).permutation(S)
    st, chp, crp, cvp, nrp = b.phase2_fused(slot[perm], rnd[perm], val[perm])
    assert st == 0 and chp.all() and (cvp == val[perm]).all()                                       # (3)
    np.testing.assert_array_equal(b.state_digest(), d1)
    b.close()

    c = fa.Context(fa.make_config(**kw))
    assert c.acceptor_phase1a(0, 2)[0] == 0
    h = S // 2
    st, c1, _, v1, _ = c.phase2_fused(slot[:h], rnd[:h], val[:h])
    stale = c.phase2_fused(slot[h:h + 4096], rnd[h:h + 4096] - 1, val[h:h + 4096] + 1)              # round 1 < 2
    assert stale[0] == 0 and not stale[1].any() and (stale[4] == 2).all()     # Nacked with the acceptors' round
    st, c2, _, v2, _ = c.phase2_fused(slot[h:], rnd[h:], val[h:])
    assert c1.all() and c2.all() and (np.concatenate([v1, v2]) == val).all()                        # (4)
    # the stale proposals left their tally entries behind (Pending forever), so the digests differ in the tallies
    # only: compare the acceptors' side
    np.testing.assert_array_equal(c.state_digest()[:5], d1[:5])
    c.close()

    st, ch5, cr5, cv5, _ = a.phase2_fused(slot, rnd + 2, val)                                       # (5) round 4
    assert st == 0 and ch5.all() and (cr5 == 4).all() and (cv5 == val).all()
    st, ch6, _, _, nr6 = a.phase2_fused(slot[:8192], rnd[:8192] + 1, val[:8192] + 7)                # round 3 < 4
    assert st == 0 and not ch6.any() and (nr6 == 4).all()
    a.close()
