"""mencius_maps.py -- TEST INFRASTRUCTURE ONLY (see fpx_oracle.h): a second, reference-shaped restatement of the Mencius
Phase-2 handlers that oracle/fpx_oracle.c restates on flat arrays, message at a time, with
the SHAPES of the Scala: one Acceptor object per (leader group, acceptor group, index) with `round` and a `states` map
(mencius/Acceptor.scala:116-119), one proxy leader with `states: Map[SlotRound, State]` keyed by
(slotStartInclusive, slotEndExclusive, round) and per-group vote maps (mencius/ProxyLeader.scala:86-110, 151).  Nothing of
the flat layout (vote arrays indexed by slot, 256-bit vote sets, the hashed range table) is shared with the code it
is held against (tests/test_mencius_models.py).  Pure-Python loops: small cases only.

    mencius/Acceptor.scala:202-235     handlePhase2a              :237-291  handlePhase2aNoopRange
    mencius/Acceptor.scala:166-196     handlePhase1a (the round it leaves behind)
    mencius/ProxyLeader.scala:216-253  handlePhase2a              :255-303  handlePhase2aNoopRange
    mencius/ProxyLeader.scala:305-353  handlePhase2b              :355-411  handlePhase2bNoopRange
    mencius/Config.scala:32-37, mencius/ProxyLeader.scala:169-176  slot -> leader group -> acceptor group
"""

NOOP = -1


class Acceptor:
    def __init__(self, leader_group, acceptor_group, index, num_leader_groups, num_acceptor_groups):
        self.lg, self.ag, self.index = leader_group, acceptor_group, index
        self.L, self.A = num_leader_groups, num_acceptor_groups
        self.round = -1                                   # Acceptor.scala:116
        self.states = {}                                  # slot -> (voteRound, voteValue)  :119

    def acceptor_group_index_by_slot(self, slot):         # :136-138
        return (slot // self.L) % self.A

    def handle_phase1a(self, round_):                     # :166-196: a stale round is Nacked, else the round moves
        if round_ < self.round:
            return ("nack", self.round)
        self.round = round_
        return ("phase1b",)

    def handle_phase2a(self, slot, round_, value):        # :202-235
        if round_ < self.round:
            return ("nack", self.round)
        self.round = round_
        self.states[slot] = (round_, value)
        return ("phase2b",)

    def handle_phase2a_noop_range(self, start, end, round_):   # :237-291
        if round_ < self.round:
            return ("nack", self.round)
        self.round = round_
        slot = start
        while self.acceptor_group_index_by_slot(slot) != self.ag:   # :263-266
            slot += self.L
        while slot < end:                                            # :268-277
            self.states[slot] = (round_, NOOP)
            slot += self.L * self.A
        return ("phase2b_noop_range",)


class ProxyLeader:
    def __init__(self, quorum_size, num_acceptor_groups):
        self.quorum_size, self.A = quorum_size, num_acceptor_groups
        self.states = {}      # (start, end, round) -> ["phase2a", value, {index}] | ["range", [{index}] * A] | "done"

    def handle_phase2a(self, slot, round_, value):        # :216-253 (the choice of the quorum is the caller's)
        key = (slot, slot + 1, round_)
        if key in self.states:
            return False
        self.states[key] = ["phase2a", value, set()]
        return True

    def handle_phase2a_noop_range(self, start, end, round_):   # :255-303
        key = (start, end, round_)
        if key in self.states:
            return False
        self.states[key] = ["range", [set() for _ in range(self.A)]]
        return True

    def handle_phase2b(self, slot, round_, acceptor_index):    # :305-353 -> None | "fatal" | ("chosen", value)
        st = self.states.get((slot, slot + 1, round_))
        if st is None:
            return "fatal"
        if st == "done" or st[0] == "range":
            return None
        st[2].add(acceptor_index)
        if len(st[2]) < self.quorum_size:
            return None
        self.states[(slot, slot + 1, round_)] = "done"
        return ("chosen", st[1])

    def handle_phase2b_noop_range(self, start, end, round_, acceptor_group, acceptor_index):   # :355-411
        st = self.states.get((start, end, round_))
        if st is None:
            return "fatal"
        if st == "done" or st[0] == "phase2a":
            return None
        st[1][acceptor_group].add(acceptor_index)
        if any(len(g) < self.quorum_size for g in st[1]):
            return None
        self.states[(start, end, round_)] = "done"
        return ("chosen_noop_range", start, end)


class Mencius:
    """every acceptor of every (leader group, acceptor group) and ONE proxy leader"""

    def __init__(self, num_leader_groups, num_acceptor_groups, acceptors_per_group, f):
        self.L, self.A, self.R = num_leader_groups, num_acceptor_groups, acceptors_per_group
        self.acceptors = [[[Acceptor(lg, ag, i, self.L, self.A) for i in range(self.R)] for ag in range(self.A)]
                          for lg in range(self.L)]
        self.proxy = ProxyLeader(f + 1, self.A)           # Config.scala:32 quorumSize = f + 1

    def group_of(self, slot):
        return slot % self.L, (slot // self.L) % self.A
