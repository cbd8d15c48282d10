"""epaxos_sets.py -- TEST INFRASTRUCTURE ONLY (see fpx_oracle.h): a second, reference-shaped restatement of the EPaxos
handlers that oracle/fpx_oracle_epaxos.c restates on flat arrays.

It keeps the SHAPES of the Scala: a cmdLog map per replica (epaxos/Replica.scala:440), CmdLogEntry case classes
(:303-330), ballots as (ordering, replicaIndex) tuples compared lexicographically (BallotHelpers.scala:11-21) -- and
dependencies as EXPLICIT SETS of instances, where the C restatement and the GPU kernels carry per-leader watermarks
plus the own-leader column's "values end".  InstancePrefixSet.fromTopOne (epaxos/InstancePrefixSet.scala:19-29) is
{(l, x) : x < topOne(l)}, subtractOne is set.discard, addAll is set union, equality is set equality: nothing of the
watermark / hole encoding is shared with the code it checks (tests/test_epaxos_models.py decodes that encoding into
sets and compares).  Pure-Python loops: small cases only.

    util/TopOne.scala:6-24, statemachine/KeyValueStore.scala:225-302   the top-one conflict index
    epaxos/Replica.scala:569-600    computeSequenceNumberAndDependencies
    epaxos/Replica.scala:633-729    transitionToPreAcceptPhase           :1159-1289  handlePreAccept
    epaxos/Replica.scala:1291-1419  handlePreAcceptOk, :796-813 preAcceptingSlowPath, :815-860 commit
    epaxos/Replica.scala:732-792    transitionToAcceptPhase              :1421-1565  handleAccept / handleAcceptOk
    epaxos/Replica.scala:1632-1757  handlePrepare                        :1759-1884  handlePrepareOk (round 5)
    epaxos/Replica.scala:1567-1575  handleCommit (round 5)
"""

NONE, NO_COMMAND, PRE_ACCEPTED, ACCEPTED, COMMITTED = range(5)
NULL_BALLOT = (-1, -1)


class Entry:
    def __init__(self, kind, ballot=NULL_BALLOT, vote_ballot=NULL_BALLOT, triple_id=-1, deps=None):
        self.kind, self.ballot, self.vote_ballot, self.triple_id = kind, ballot, vote_ballot, triple_id
        self.deps = deps        # a frozenset of (leader, number), or None: the triple is known by its id only


class Replica:
    def __init__(self, index, n, num_keys):
        self.index, self.n = index, n
        self.cmd_log = {}                                   # Instance -> Entry
        self.largest_ballot = (0, index)                    # Replica.scala:458
        self.gets = [[0] * n for _ in range(num_keys)]      # KeyValueStore.scala:229-230: TopOne per key
        self.sets = [[0] * n for _ in range(num_keys)]

    # KeyValueStore.getTopOneConflicts for a single-key command (:259-302): a get conflicts with the sets of its
    # key, a set with its gets and sets; TopOne.mergeEquals is an elementwise max
    def top_one_conflicts(self, key, is_set):
        merged = list(self.sets[key])
        if is_set:
            merged = [max(a, b) for a, b in zip(merged, self.gets[key])]
        return merged

    # KeyValueStore.put (:232-253) -> TopOne.put (util/TopOne.scala:12-15)
    def index_put(self, key, is_set, instance):
        leader, number = instance
        row = (self.sets if is_set else self.gets)[key]
        row[leader] = max(row[leader], number + 1)

    # computeSequenceNumberAndDependencies (:569-600): InstancePrefixSet.fromTopOne(conflicts) minus the instance
    def compute_dependencies(self, instance, key, is_set):
        if key < 0:                                         # Noop: InstancePrefixSet(n), empty (:592-593)
            return set()
        top = self.top_one_conflicts(key, is_set)
        deps = {(l, x) for l in range(self.n) for x in range(top[l])}
        deps.discard(instance)                              # dependencies.subtractOne(instance) :582
        return deps


class EPaxos:
    def __init__(self, n, num_keys):
        self.n, self.f = n, (n - 1) // 2
        self.replicas = [Replica(r, n, num_keys) for r in range(n)]

    # ---- one tick of fresh instances (the tick-at-once form of fpo_epx_preaccept) ---------------------------
    def tick(self, leader, number, key, is_set, resp_mask, rank, seen_mask=None, triple_id=None):
        """returns per message (fast, deps, leader_deps) with deps as sets"""
        m, n = len(leader), self.n
        seen_mask = resp_mask if seen_mask is None else seen_mask
        local = [[None] * n for _ in range(m)]
        for r, rep in enumerate(self.replicas):
            order = sorted(range(m), key=lambda i: rank[r][i])
            for i in order:
                inst = (int(leader[i]), int(number[i]))
                if r != inst[0] and not (int(seen_mask[i]) >> r) & 1:
                    continue
                assert inst not in rep.cmd_log             # cmdLog.get(instance) == None: the only branch of a tick
                local[i][r] = rep.compute_dependencies(inst, int(key[i]), bool(is_set[i]))
                rep.index_put(int(key[i]), bool(is_set[i]), inst)      # updateConflictIndex :696 / :1279
        out = []
        for i in range(m):
            L = int(leader[i])
            inst = (L, int(number[i]))
            D = local[i][L]                                # transitionToPreAcceptPhase :641-642
            answers = {}
            for r in range(n):
                if r != L and (int(seen_mask[i]) >> r) & 1:
                    answers[r] = local[i][r] | D           # handlePreAccept :1257-1262
            first_quorum = [answers[r] for r in range(n) if (int(resp_mask[i]) >> r) & 1]
            # handlePreAcceptOk :1376-1410: popularItems(the n-2 other answers, n-2) non-empty iff they all agree
            fast = all(a == first_quorum[0] for a in first_quorum)
            deps = first_quorum[0] if fast else set().union(D, *first_quorum)   # preAcceptingSlowPath :796-813
            tid = -1 if triple_id is None else int(triple_id[i])
            for r, rep in enumerate(self.replicas):
                if fast:                                   # commit :815-823 + Commit to the others
                    rep.cmd_log[inst] = Entry(COMMITTED, triple_id=tid, deps=frozenset(deps))
                elif r == L:
                    rep.cmd_log[inst] = Entry(PRE_ACCEPTED, (0, L), (0, L), tid, frozenset(D))       # :688-696
                elif r in answers:
                    rep.cmd_log[inst] = Entry(PRE_ACCEPTED, (0, L), (0, L), tid, frozenset(answers[r]))  # :1265-1276
            out.append((fast, deps, D))
        for rep in self.replicas:                           # commit -> updateConflictIndex reaches every replica :815-828
            for i in range(m):
                rep.index_put(int(key[i]), bool(is_set[i]), (int(leader[i]), int(number[i])))
        return out

    # ---- Replica.handlePreAccept, every branch (:1159-1289), message at a time ------------------------------
    def handle_preaccept(self, instance, ballot, key, is_set, triple_id, deps_in, targets):
        """returns {replica: ('ok' | 'resend' | 'commit', deps or None, triple id) | ('nack', largestBallot) | ('ignore',)}"""
        replies = {}
        for r in targets:
            rep = self.replicas[r]
            e = rep.cmd_log.get(instance)
            if e is not None:
                if e.kind == COMMITTED:                                               # :1227-1238
                    replies[r] = ("commit", e.deps, e.triple_id)
                    continue
                if ballot < e.ballot:                                                 # :1180, :1189, :1215
                    replies[r] = ("nack", rep.largest_ballot)
                    continue
                if e.kind == PRE_ACCEPTED and ballot == e.vote_ballot:                # :1196-1210
                    replies[r] = ("resend", e.deps, e.triple_id)
                    continue
                if e.kind == ACCEPTED and ballot == e.vote_ballot:                    # :1222-1224
                    replies[r] = ("ignore",)
                    continue
            rep.largest_ballot = max(rep.largest_ballot, ballot)                      # :1251
            deps = rep.compute_dependencies(instance, key, is_set) | set(deps_in)     # :1255-1262
            rep.cmd_log[instance] = Entry(PRE_ACCEPTED, ballot, ballot, triple_id, frozenset(deps))   # :1265-1276
            if key >= 0:
                rep.index_put(key, is_set, instance)                                  # :1279 (a Noop leaves the index alone)
            replies[r] = ("ok", frozenset(deps), triple_id)
        return replies

    # ---- Replica.handlePrepare (:1632-1757) -----------------------------------------------------------------
    def prepare(self, instance, ballot, targets):
        replies = {}
        for r in targets:
            rep = self.replicas[r]
            rep.largest_ballot = max(rep.largest_ballot, ballot)                      # :1637
            e = rep.cmd_log.get(instance)
            if e is not None and e.kind == COMMITTED:
                replies[r] = ("commit",)                                              # :1744-1755
            elif e is not None and ballot < e.ballot:
                replies[r] = ("nack", rep.largest_ballot)
            elif e is None or e.kind == NO_COMMAND:                                   # :1654-1669, :1686-1701
                rep.cmd_log[instance] = Entry(NO_COMMAND, ballot)
                replies[r] = ("ok", NONE, NULL_BALLOT, -1)
            else:                                                                     # :1711-1743: only `ballot` moves
                e.ballot = ballot
                replies[r] = ("ok", e.kind, e.vote_ballot, e.triple_id)
        return replies

    # ---- the Accept phase of one instance (:732-792, 1421-1565, 815-860) ------------------------------------
    def accept(self, instance, ballot, triple_id, targets, key=-1, is_set=False):
        """returns (fatal, replies, committed); key / is_set: the triple's command (-1 = Noop: updateConflictIndex
        has no bytes to put, Replica.scala:602-614)"""
        P = ballot[1]
        prop = self.replicas[P]
        put = (lambda rep: rep.index_put(key, is_set, instance)) if key >= 0 else (lambda rep: None)
        e = prop.cmd_log.get(instance)
        if e is not None and (e.kind == COMMITTED or e.ballot > ballot or
                              (e.kind in (PRE_ACCEPTED, ACCEPTED) and e.vote_ballot > ballot)):
            return True, {}, False                                                    # logger.fatal / checkLe :740-757
        prop.cmd_log[instance] = Entry(ACCEPTED, ballot, ballot, triple_id, None)     # :759-762
        put(prop)                                                                     # :763
        replies = {P: ("ok",)}                                                        # its own AcceptOk :780-789
        for r in targets:
            rep = self.replicas[r]
            e = rep.cmd_log.get(instance)
            if e is not None and e.kind == COMMITTED:
                replies[r] = ("commit",)                                              # :1463-1474
            elif e is not None and ballot < e.ballot:
                replies[r] = ("nack", rep.largest_ballot)                             # :1432-1449
            elif e is not None and e.kind == ACCEPTED and ballot == e.vote_ballot:
                replies[r] = ("ok",)                                                  # :1451-1461 re-sent AcceptOk
            else:
                rep.largest_ballot = max(rep.largest_ballot, ballot)                  # :1487
                rep.cmd_log[instance] = Entry(ACCEPTED, ballot, ballot, triple_id, None)   # :1493-1502
                put(rep)                                                              # :1503
                replies[r] = ("ok",)
        oks = [r for r, v in replies.items() if v[0] == "ok"]
        committed = len(oks) >= self.f + 1                                            # slowQuorumSize, :1557-1563
        if committed:
            for rep in self.replicas:                                                 # commit + Commit to the others
                rep.cmd_log[instance] = Entry(COMMITTED, triple_id=triple_id, deps=None)
                put(rep)                                                              # commit :828
        return False, replies, committed

    # ---- Replica.handleCommit (:1567-1575) -> commit (:815-830) at the replicas of `targets` ----------------------
    def handle_commit(self, instance, triple_id, deps, targets, key=-1, is_set=False):
        """deps: the Commit's dependencies as a set of instances, or None (the triple is known by its id alone)"""
        for r in targets:
            rep = self.replicas[r]
            # cmdLog -= instance; cmdLog(instance) = CommittedEntry(triple) (:826-827): no ballot is looked at
            rep.cmd_log[instance] = Entry(COMMITTED, triple_id=triple_id, deps=None if deps is None else frozenset(deps))
            if key >= 0:
                rep.index_put(key, is_set, instance)                                  # updateConflictIndex :828

    # ---- Replica.handlePrepareOk (:1759-1884): what the recovering replica `me` decides once `responses` are in ----
    def handle_prepare_oks(self, instance, ballot, me, responses, as_intended=False):
        """responses: {replica: (status, vote_ballot, triple_id, deps)} -- PrepareOk.status as an entry kind (NONE =
        CommandStatus.NotSeen), voteBallot as a tuple, the triple as (id, frozenset of dependencies or None).
        Returns ("wait",) | ("accept", replica, triple_id) | ("preaccept", replica, triple_id) | ("noop",).

        as_intended = False restates the handler AS SCALA EVALUATES IT:
          :1810  prepareOks.find(_.status == Some(CommandStatus.Accepted)) compares a CommandStatus with an Option: never true;
          :1831  .filter(p => p.ballot == Ballot(0, p.instance.replicaIndex)) reads PrepareOk.ballot -- the ballot of the
                 Prepare being answered, the same for every response -- not the ballot the vote was cast in.
        as_intended = True is what the comments beside them describe (:1806-1809, :1826-1829)."""
        if len(responses) < self.f + 1:                                               # :1799 slowQuorumSize
            return ("wait",)
        max_ballot = max(v[1] for v in responses.values())                            # :1805 (BallotHelpers.Ordering)
        oks = {r: v for r, v in responses.items() if v[1] == max_ballot}              # :1806
        if as_intended:
            for r in sorted(oks):                                                     # :1810-1824 "if some response was accepted"
                if oks[r][0] == ACCEPTED:
                    return ("accept", r, oks[r][2])
        # :1830-1843 PreAccepted responses in the default ballot, not from the recovering replica itself
        default = (0, instance[0])
        in_default = (max_ballot == default) if as_intended else (ballot == default)
        cands = [(r, v) for r, v in sorted(oks.items()) if v[0] == PRE_ACCEPTED and in_default and r != me]
        for r, v in cands:                                                            # Util.popularItems(.., config.f): :1844-1851
            same = sum(1 for _, w in cands if (w[2], w[3]) == (v[2], v[3]))          # CommandTriple equality: command + dependencies
            if same >= self.f:
                return ("accept", r, v[2])
        for r in sorted(oks):                                                         # :1856-1868
            if oks[r][0] == PRE_ACCEPTED:
                return ("preaccept", r, oks[r][2])
        return ("noop",)
