"""Two independent restatements of the EPaxos handlers must agree: oracle/fpx_oracle_epaxos.c (flat arrays, per-leader
watermarks + the own column's "values end" -- the encoding the GPU kernels share) against oracle/epaxos_sets.py
(reference-shaped: cmdLog maps, ballot tuples, dependencies as explicit sets of instances).  The encoded answers of
the first are decoded into sets and compared with the second on random scenarios that mix pre-accept ticks, general
PreAccepts (ballots up and down, Noops, holes), Prepares and Accepts.  CPU only."""
import numpy as np
import pytest

from oracle import epaxos_sets as model


COVERAGE = []


def decode(watermarks, own_leader, number, values_end):
    """the instance set the (watermarks, values_end) encoding of an instance (own_leader, number) stands for"""
    s = {(l, y) for l, w in enumerate(watermarks) for y in range(int(w))}
    if values_end:
        assert int(watermarks[own_leader]) == number
        s |= {(own_leader, y) for y in range(number + 1, int(values_end))}
    return s


def encode_ballot(b):
    return -1 if b[0] < 0 else b[0] * 8 + b[1]


def compare_state(oracle_e, mod, n, NI, num_keys):
    for r in range(n):
        rep = mod.replicas[r]
        for L in range(n):
            for x in range(NI):
                kind, ballot, vote, tid, largest = oracle_e.read_cmdlog(r, L, x)
                e = rep.cmd_log.get((L, x))
                assert largest == encode_ballot(rep.largest_ballot)
                if e is None:
                    assert kind == 0
                    continue
                assert (kind, ballot, vote, tid) == (e.kind, encode_ballot(e.ballot), encode_ballot(e.vote_ballot), e.triple_id), (r, L, x)
                if e.kind in (model.PRE_ACCEPTED, model.COMMITTED, model.ACCEPTED):
                    wm, end = oracle_e.read_cmdlog_deps(r, L, x)
                    if e.deps is None:
                        assert wm[0] == -1
                    else:
                        assert decode(wm, L, x, end) == set(e.deps), (r, L, x)
        for k in range(num_keys):
            g, s = oracle_e.read_index(r, k)
            assert g.tolist() == rep.gets[k] and s.tolist() == rep.sets[k]


@pytest.mark.parametrize("n,num_keys,seed", [(3, 2, 1), (5, 3, 2), (5, 2, 3), (7, 3, 4), (5, 1, 5), (3, 3, 6)])
def test_flat_oracle_and_set_model_agree(oracle, n, num_keys, seed):
    NI = 24
    rng = np.random.default_rng(seed)
    ref = oracle.EPaxos(n, num_keys, num_instances=NI)
    mod = model.EPaxos(n, num_keys)
    nxt = [0] * n
    seen = dict(ok=0, resend=0, nack=0, commit=0, ignore=0, hole=0, fatal=0, committed=0, fast=0, slow=0)
    for step in range(240):
        kind = int(rng.integers(0, 4))
        if kind == 0 and max(nxt) < NI // 2 - 3:
            # a tick of fresh instances with per-replica delivery orders (channels may reorder: holes)
            m = int(rng.integers(1, 7))
            leader = rng.integers(0, n, m).astype(np.int32)
            number = np.zeros(m, np.int32)
            for i in range(m):
                number[i] = nxt[leader[i]]
                nxt[leader[i]] += 1
            key = rng.integers(0, num_keys, m).astype(np.int32)
            is_set = rng.integers(0, 2, m).astype(np.uint8)
            mask = np.zeros(m, np.uint8)
            seenm = np.zeros(m, np.uint8)
            for i in range(m):
                others = [r for r in range(n) if r != leader[i]]
                quorum = rng.choice(others, size=n - 2, replace=False)
                mask[i] = sum(1 << int(r) for r in quorum)
                seenm[i] = mask[i] | (sum(1 << r for r in others) if rng.random() < 0.5 else 0)
            rank = np.stack([rng.permutation(m) for _ in range(n)]).astype(np.int32)
            tr = rng.integers(0, 1000, m).astype(np.int32)
            # the leader numbers its instances in ITS order: make every leader's own rank increasing in the number
            for L in range(n):
                idx = np.nonzero(leader == L)[0]
                rank[L, idx] = np.sort(rank[L, idx])
            st, fast, deps, ldeps, own = ref.preaccept(leader, number, key, is_set, mask, rank, seen_mask=seenm, triple_id=tr)
            assert st == 0
            out = mod.tick(leader, number, key, is_set, mask, rank, seen_mask=seenm, triple_id=tr)
            for i in range(m):
                f, d, D = out[i]
                assert bool(fast[i]) == f
                assert decode(deps[i], int(leader[i]), int(number[i]), own[i][0]) == d, (step, i)
                assert decode(ldeps[i], int(leader[i]), int(number[i]), own[i][1]) == D
                seen["fast" if f else "slow"] += 1
                seen["hole"] += int(own[i][0] != 0)
            continue
        # one message about an instance in any state: known ones, or one from the upper half of the numbers
        L = int(rng.integers(0, n))
        x = int(rng.integers(0, max(1, nxt[L]))) if rng.random() < 0.7 and nxt[L] else int(rng.integers(NI // 2, NI))
        ballot = (int(rng.integers(0, 3)), int(rng.integers(0, n)))
        if rng.random() < 0.3:
            ballot = (0, L)                                  # the original leader's default ballot
        elif rng.random() < 0.3:                             # a ballot some replica has voted in for this instance
            votes = [rep.cmd_log[(L, x)].vote_ballot for rep in mod.replicas
                     if (L, x) in rep.cmd_log and rep.cmd_log[(L, x)].kind in (model.PRE_ACCEPTED, model.ACCEPTED)]
            if votes:
                ballot = votes[int(rng.integers(0, len(votes)))]
        targets = [r for r in range(n) if rng.random() < 0.6]
        tmask = [sum(1 << r for r in targets)]
        if kind == 1:
            key = int(rng.integers(-1, num_keys))
            is_set = int(rng.integers(0, 2))
            tid = int(rng.integers(0, 1000))
            din = rng.integers(0, NI, n).astype(np.int32)
            hole = rng.random() < 0.4
            din[L] = x if hole else min(int(din[L]), x)
            dend = x + 2 + int(rng.integers(0, 4)) if hole else 0
            st, ok, resend, nack, com, nb, rd, re, rt = ref.handle_preaccept([L], [x], [ballot[0]], [ballot[1]], [key], [is_set],
                                                                            [tid], [din], [dend], tmask)
            assert st == 0
            got = mod.handle_preaccept((L, x), ballot, key, bool(is_set), tid, decode(din, L, x, dend), targets)
            nacks = []
            for r in range(n):
                bit = 1 << r
                v = got.get(r)
                if v is None or v[0] == "ignore":
                    assert not ((ok[0] | resend[0] | nack[0] | com[0]) & bit)
                    seen["ignore"] += v is not None
                    continue
                field = dict(ok=ok, resend=resend, nack=nack, commit=com)[v[0]]
                assert field[0] & bit, (step, r, v[0])
                seen[v[0]] += 1
                if v[0] == "nack":
                    nacks.append(v[1])
                else:
                    assert rt[0][r] == v[2]
                    if v[1] is None:
                        assert rd[0][r][0] == -1
                    else:
                        assert decode(rd[0][r], L, x, re[0][r]) == set(v[1]), (step, r)
                        seen["hole"] += int(re[0][r] != 0)
            assert nb[0] == (encode_ballot(max(nacks)) if nacks else -1)
        elif kind == 2:
            st, ok, nack, com, nb, rs, rv, rt = ref.prepare([L], [x], [ballot[0]], [ballot[1]], tmask)
            assert st == 0
            got = mod.prepare((L, x), ballot, targets)
            nacks = []
            for r in range(n):
                bit, v = 1 << r, got.get(r)
                if v is None:
                    assert not ((ok[0] | nack[0] | com[0]) & bit) and rs[0][r] == -1
                elif v[0] == "commit":
                    assert com[0] & bit
                elif v[0] == "nack":
                    assert nack[0] & bit
                    nacks.append(v[1])
                else:
                    assert ok[0] & bit and (rs[0][r], rv[0][r], rt[0][r]) == (v[1], encode_ballot(v[2]), v[3])
            assert nb[0] == (encode_ballot(max(nacks)) if nacks else -1)
        else:
            targets = [r for r in targets if r != ballot[1]]
            tmask = [sum(1 << r for r in targets)]
            tid = int(rng.integers(0, 1000))
            akey, aset = int(rng.integers(-1, num_keys)), bool(rng.integers(0, 2))   # the triple's command; -1 = Noop
            st, ok, nack, com, nb, done = ref.accept([L], [x], [ballot[0]], [ballot[1]], [tid], tmask, [akey], [aset])
            fatal, got, committed = mod.accept((L, x), ballot, tid, targets, akey, aset)
            assert (st == 9) == fatal and st in (0, 9)
            seen["fatal"] += fatal
            if not fatal:
                assert bool(done[0]) == committed
                seen["committed"] += committed
                for r in range(n):
                    bit, v = 1 << r, got.get(r)
                    word = (v or ("none",))[0]
                    assert bool(ok[0] & bit) == (word == "ok") and bool(nack[0] & bit) == (word == "nack")
                    assert bool(com[0] & bit) == (word == "commit")
        if step % 40 == 39:
            compare_state(ref, mod, n, NI, num_keys)
    compare_state(ref, mod, n, NI, num_keys)
    COVERAGE.append(seen)
    assert seen["ok"] and seen["fast"] + seen["slow"] > 0, seen


@pytest.mark.parametrize("n", [3, 5, 7])
@pytest.mark.parametrize("as_intended", [False, True])
def test_recovery_decisions_and_commits_from_outside_agree(oracle, n, as_intended):
    total = dict(wait=0, accept=0, preaccept=0, noop=0, commits=0)
    for seed in range(11, 17):
        for k, v in _recovery_scenario(oracle, n, seed, as_intended).items():
            total[k] += v
    # every decision was reached.  (As written, "Accept phase" needs the recovering replica to be the original leader in its
    # default ballot -- :1831 reads the Prepare's ballot -- with f matching PreAccepted answers from others: rare, and never
    # at n = 3 in these histories; in the intended reading an Accepted response wins: :1810)
    assert total["commits"] and total["wait"] and total["noop"] and total["preaccept"], total
    assert total["accept"] or (not as_intended and n == 3), total


def _recovery_scenario(oracle, n, seed, as_intended):
    """round 5: the two handlers that had no second restatement.  Replica.handleCommit (:1567-1575) --
    Commits for instances in every state, with dependencies or by triple id -- and Replica.handlePrepareOk (:1759-1884) in
    BOTH readings (as Scala evaluates :1810 / :1831, and as the comments beside them intend): random histories of ticks,
    PreAccepts in higher ballots, Accepts and Commits build command logs; then a recovering replica's Prepare collects
    PrepareOks from a random quorum and both restatements decide -- wait / Accept phase with a triple / pre-accept again /
    Noop, the same source replica, the same triple."""
    NI, num_keys = 24, 2
    rng = np.random.default_rng(seed)
    ref = oracle.EPaxos(n, num_keys, num_instances=NI)
    mod = model.EPaxos(n, num_keys)
    f = (n - 1) // 2
    nxt = [0] * n
    seen = dict(wait=0, accept=0, preaccept=0, noop=0, commits=0)
    for step in range(160):
        kind = int(rng.integers(0, 5))
        if kind == 0 and max(nxt) < NI // 2 - 3:
            m = int(rng.integers(1, 5))
            leader = rng.integers(0, n, m).astype(np.int32)
            number = np.zeros(m, np.int32)
            for i in range(m):
                number[i] = nxt[leader[i]]
                nxt[leader[i]] += 1
            key = rng.integers(0, num_keys, m).astype(np.int32)
            is_set = rng.integers(0, 2, m).astype(np.uint8)
            mask = np.zeros(m, np.uint8)
            for i in range(m):
                others = [r for r in range(n) if r != leader[i]]
                mask[i] = sum(1 << int(r) for r in rng.choice(others, size=n - 2, replace=False))
            rank = np.stack([rng.permutation(m) for _ in range(n)]).astype(np.int32)
            for L in range(n):
                idx = np.nonzero(leader == L)[0]
                rank[L, idx] = np.sort(rank[L, idx])
            tr = rng.integers(0, 1000, m).astype(np.int32)
            assert ref.preaccept(leader, number, key, is_set, mask, rank, triple_id=tr)[0] == 0
            mod.tick(leader, number, key, is_set, mask, rank, triple_id=tr)
            continue
        L = int(rng.integers(0, n))
        x = int(rng.integers(0, max(1, nxt[L]))) if rng.random() < 0.3 and nxt[L] else int(rng.integers(NI // 2, NI // 2 + 3))
        if kind == 1 and rng.random() < 0.7:
            x = int(rng.integers(NI // 2 + 3, NI))          # (most Commits go elsewhere: the pool above stays recoverable)
        targets = [r for r in range(n) if rng.random() < 0.6]
        tmask = [sum(1 << r for r in targets)]
        if kind == 1:       # a Commit from outside
            key, is_set, tid = int(rng.integers(-1, num_keys)), int(rng.integers(0, 2)), int(rng.integers(0, 1000))
            if rng.random() < 0.3:
                assert ref.handle_commit([L], [x], [tid], tmask, key=[key], is_set=[is_set]) == 0
                mod.handle_commit((L, x), tid, None, targets, key, bool(is_set))
            else:
                din = rng.integers(0, NI, n).astype(np.int32)
                hole = rng.random() < 0.4
                din[L] = x if hole else min(int(din[L]), x)
                dend = x + 2 + int(rng.integers(0, 4)) if hole else 0
                assert ref.handle_commit([L], [x], [tid], tmask, key=[key], is_set=[is_set], deps=[din], deps_values_end=[dend]) == 0
                mod.handle_commit((L, x), tid, decode(din, L, x, dend), targets, key, bool(is_set))
            seen["commits"] += 1
        elif kind == 2:     # a PreAccept in some ballot (re-sent, or a recovering replica's): entries with other vote ballots
            ballot = (int(rng.integers(0, 3)), int(rng.integers(0, n)))
            key, is_set, tid = int(rng.integers(-1, num_keys)), int(rng.integers(0, 2)), int(rng.integers(0, 1000))
            din = rng.integers(0, NI, n).astype(np.int32)
            din[L] = min(int(din[L]), x)
            assert ref.handle_preaccept([L], [x], [ballot[0]], [ballot[1]], [key], [is_set], [tid], [din], [0], tmask)[0] == 0
            mod.handle_preaccept((L, x), ballot, key, bool(is_set), tid, decode(din, L, x, 0), targets)
        elif kind == 3:     # an Accept
            ballot = (int(rng.integers(0, 3)), int(rng.integers(0, n)))
            t2 = [r for r in targets if r != ballot[1]][:max(0, f - 1)]      # short of a quorum: the entries stay Accepted
            tid = int(rng.integers(0, 1000))
            st = ref.accept([L], [x], [ballot[0]], [ballot[1]], [tid], [sum(1 << r for r in t2)], [-1], [0])[0]
            fatal = mod.accept((L, x), ballot, tid, t2, -1, False)[0]
            assert (st == 9) == fatal
        else:               # a recovery: Prepare in a ballot of `me` to a random set of replicas, then the decision
            ballot = (int(rng.integers(0, 4)), int(rng.integers(0, n))) if rng.random() < 0.8 else (0, L)
            me = ballot[1]                                    # a replica recovers in a ballot of its own (:1001-1019)
            st, ok, nack, com, nb, rs, rv, rt = ref.prepare([L], [x], [ballot[0]], [ballot[1]], tmask)
            assert st == 0
            got = mod.prepare((L, x), ballot, targets)
            resp = {r: v for r, v in got.items() if v[0] == "ok"}
            for r, v in resp.items():
                assert ok[0] & (1 << r) and (rs[0][r], rv[0][r], rt[0][r]) == (v[1], encode_ballot(v[2]), v[3])
            rmask = [sum(1 << r for r in resp)]
            st, act, src, tr = ref.handle_prepare_oks([L], [x], [ballot[0]], [ballot[1]], rmask, rs, rv, rt, as_intended=as_intended)
            assert st == 0
            # the set model's responses carry the triples' dependencies as sets (what the reply's InstancePrefixSet holds)
            full = {}
            for r, v in resp.items():
                e = mod.replicas[r].cmd_log.get((L, x))
                full[r] = (v[1], v[2], v[3], e.deps if e is not None and v[1] in (model.PRE_ACCEPTED, model.ACCEPTED) else None)
            want = mod.handle_prepare_oks((L, x), ballot, me, full, as_intended=as_intended)
            word = {0: "wait", 1: "accept", 2: "preaccept", 3: "noop"}[int(act[0])]
            assert word == want[0], (step, word, want)
            if word in ("accept", "preaccept"):
                assert (int(src[0]), int(tr[0])) == (want[1], want[2]), (step, want)
            seen[word] += 1
    compare_state(ref, mod, n, NI, num_keys)
    return seen


def test_decode_is_the_int_prefix_set_the_reference_builds(oracle):
    """decode() against the IntPrefixSet restatement pinned on the reference's own tests (tests/test_epaxos.py):
    watermark w and subtractOne(x) with x < w is watermark x + values x+1 .. w-1"""
    import ctypes as C
    L = oracle.lib()
    out = np.zeros(64, np.int32)
    wm = C.c_int(0)
    for w in range(0, 9):
        for x in range(0, 9):
            k = L.fpo_ips_subtract_one(w, None, 0, x, out.ctypes.data_as(C.POINTER(C.c_int)), C.byref(wm))
            members = set(range(wm.value)) | set(out[:k].tolist())
            assert members == set(range(w)) - {x}
            end = (out[k - 1] + 1) if k else 0
            assert decode([wm.value], 0, x, end) == {(0, y) for y in members}


def test_the_scenarios_reached_every_branch():
    """(runs after the scenarios above) every reply kind, holes, fatal proposers, commits, both pre-accept paths"""
    total = {k: sum(c[k] for c in COVERAGE) for k in COVERAGE[0]} if COVERAGE else {}
    assert COVERAGE and all(v > 0 for v in total.values()), sorted(total.items())
