"""multipaxos_maps.py -- TEST INFRASTRUCTURE ONLY (see fpx_oracle.h): a second, reference-shaped restatement of the
MultiPaxos Phase-2 handlers that oracle/fpx_oracle.c restates on flat arrays, message at
a time, with the SHAPES of the Scala: one Acceptor object per (group, index) with `round`, a `states` map and
`maxVotedSlot` (multipaxos/Acceptor.scala:95-104), a proxy leader with `states: Map[SlotRound, State]` whose Pending
holds `phase2bs: Map[(group, acceptor), Phase2b]` (multipaxos/ProxyLeader.scala:83-135), quorum systems as sets of
nodes (quorums/{SimpleMajority, Grid, UnanimousWrites}.scala).  oracle/fpx_faithful.cpp is the C++ sibling of this file
for the steady stream (it is also the "faithful shapes" CPU baseline); this one follows every branch -- Nacks,
re-proposals, duplicates, flexible quorums -- and is held against the flat oracle by tests/test_multipaxos_models.py.
Pure-Python loops: small cases only.

    multipaxos/Acceptor.scala:148-182  handlePhase1a        :184-220  handlePhase2a
    multipaxos/ProxyLeader.scala:175-215  handlePhase2a     :217-258  handlePhase2b
    quorums/SimpleMajority.scala:41-49, quorums/Grid.scala:43-50, quorums/UnanimousWrites.scala:44-51
    multipaxos/Leader.scala:306-329 maxPhase1bSlot / safeValue, :543-566 the recovery loop over chosenWatermark..maxSlot
"""


class Acceptor:
    def __init__(self, group, index):
        self.group, self.index = group, index
        self.round = -1                 # Acceptor.scala:95
        self.states = {}                # slot -> (voteRound, voteValue)  :98
        self.max_voted_slot = -1        # :104

    def handle_phase1a(self, round_):   # :148-182
        if round_ < self.round:
            return ("nack", self.round)
        self.round = round_
        return ("phase1b",)

    def handle_phase2a(self, slot, round_, value):   # :184-220 -- `<`, not `<=`: an equal round votes again
        if round_ < self.round:
            return ("nack", self.round)
        self.round = round_
        self.states[slot] = (round_, value)
        self.max_voted_slot = max(self.max_voted_slot, slot)
        return ("phase2b",)


class SimpleMajority:                   # quorums/SimpleMajority.scala
    def __init__(self, members):
        self.members = set(members)

    def is_write_quorum(self, xs):
        assert set(xs) <= self.members  # require(xs.subsetOf(members))
        return len(set(xs)) >= len(self.members) // 2 + 1


class UnanimousWrites:                  # quorums/UnanimousWrites.scala
    def __init__(self, members):
        self.members = set(members)

    def is_write_quorum(self, xs):
        assert set(xs) <= self.members
        return set(xs) == self.members


class Grid:                             # quorums/Grid.scala: a write quorum meets every row
    def __init__(self, rows):
        self.rows = [set(r) for r in rows]
        self.nodes = set().union(*self.rows)

    def is_write_quorum(self, xs):
        assert set(xs) <= self.nodes
        return all(any(x in xs for x in row) for row in self.rows)


class ProxyLeader:
    def __init__(self, f, quorum_system_of_group=None):
        self.f = f
        self.quorum_system_of_group = quorum_system_of_group   # None: non-flexible, f + 1 votes (:238)
        self.states = {}                # (slot, round) -> ["pending", value, {(group, index)}] | "done"

    def handle_phase2a(self, slot, round_, value):   # :175-215 (which acceptors get it is the caller's choice)
        if (slot, round_) in self.states:
            return False
        self.states[(slot, round_)] = ["pending", value, set()]
        return True

    def handle_phase2b(self, slot, round_, group, index):   # :217-258 -> "fatal" | None | ("chosen", value)
        st = self.states.get((slot, round_))
        if st is None:
            return "fatal"
        if st == "done":
            return None
        st[2].add((group, index))
        if self.quorum_system_of_group is None:
            if len(st[2]) < self.f + 1:
                return None
        elif not self.quorum_system_of_group(group).is_write_quorum({i for (_, i) in st[2]}):
            return None
        self.states[(slot, round_)] = "done"
        return ("chosen", st[1])


NOOP = -1


def phase1b(acceptor, chosen_watermark):
    """Acceptor.handlePhase1a's reply (Acceptor.scala:167-181): the votes at or above the watermark"""
    return {slot: st for slot, st in acceptor.states.items() if slot >= chosen_watermark}


def leader_recovery(phase1bs_by_group, chosen_watermark, num_groups):
    """Leader.scala:543-566: maxSlot over every Phase1b received; for slot in chosenWatermark..maxSlot the safe value
    among the Phase1b's of the slot's acceptor group (:318-329: the vote with the highest voteRound, Noop if none).
    phase1bs_by_group[g] = the Phase1b infos ({slot: (voteRound, voteValue)}) of the acceptors that answered.
    Returns (maxSlot, [(voteRound or -1, value or NOOP)])."""
    max_slot = max([max(info) if info else -1 for group in phase1bs_by_group for info in group] or [-1])   # :306-312
    out = []
    for slot in range(chosen_watermark, max_slot + 1):
        infos = [info[slot] for info in phase1bs_by_group[slot % num_groups] if slot in info]
        out.append(max(infos, key=lambda rv: rv[0]) if infos else (-1, NOOP))
    return max_slot, out
