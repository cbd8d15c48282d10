// fpx_faithful.cpp -- TEST INFRASTRUCTURE ONLY: CPU baseline "B1 faithful" of BASELINE.md section 2.
//
// The same handlers as fpx_oracle.c, but with the DATA-STRUCTURE SHAPES of the Scala reference, to
// approximate what its JVM in-process simulation pays per message (it is still C++, not the JVM):
//   Acceptor.states      mutable.SortedMap[Slot, State]        -> std::map<int, State>          (Acceptor.scala:98)
//   ProxyLeader.states   mutable.Map[SlotRound, State]         -> std::unordered_map            (ProxyLeader.scala:135)
//   Pending.phase2bs     mutable.Map[(Group, Acceptor), Phase2b] -> std::map<std::pair<int,int>, Phase2b> (:93-96)
//   FakeTransport        Buffer of (src, dst, bytes), one heap object per message, drained FIFO  (FakeTransport.scala:89-159)
// Single-threaded by the Transport contract (Transport.scala:37-39).
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct State { int voteRound; int voteValue; };
struct Phase2a { int slot, round, value; };
struct Phase2b { int groupIndex, acceptorIndex, slot, round; };

struct Acceptor {                 // multipaxos/Acceptor.scala
  int round = -1;                 // :95
  std::map<int, State> states;    // :98
  int maxVotedSlot = -1;          // :104
};

struct SlotRound { int slot, round; bool operator==(const SlotRound& o) const { return slot == o.slot && round == o.round; } };
struct SlotRoundHash { size_t operator()(const SlotRound& k) const { return std::hash<int64_t>()(((int64_t)k.slot << 32) ^ (uint32_t)k.round); } };
struct Pending { Phase2a phase2a; std::map<std::pair<int, int>, Phase2b> phase2bs; };
struct PlState { bool done = false; std::unique_ptr<Pending> pending; };

struct Msg { int kind; int acceptor; Phase2a p2a; Phase2b p2b; };  // 0: Phase2a -> proxy leader, 1: Phase2a -> acceptor, 2: Phase2b

}  // namespace

// Runs `slots` Phase2a's (round 0, value = value[i]) through proxy leader + R acceptors (threshold
// quorum f+1, dense delivery) with a FIFO message pump; returns the number of Chosen messages and
// a checksum of the chosen values.
extern "C" int64_t fpo_faithful_run(int32_t slots, int32_t R, int32_t f, const int32_t* value, int64_t* checksum) {
  std::vector<Acceptor> acceptors((size_t)R);
  for (auto& a : acceptors) a.round = 0;  // the leader's Phase 1 in round 0
  std::unordered_map<SlotRound, PlState, SlotRoundHash> plStates;
  std::deque<std::unique_ptr<Msg>> transport;
  int64_t chosen = 0, sum = 0;
  for (int s = 0; s < slots; ++s) {
    auto m = std::make_unique<Msg>();
    m->kind = 0;
    m->p2a = Phase2a{s, 0, value[s]};
    transport.push_back(std::move(m));
  }
  while (!transport.empty()) {
    std::unique_ptr<Msg> m = std::move(transport.front());
    transport.pop_front();
    if (m->kind == 0) {  // ProxyLeader.handlePhase2a, ProxyLeader.scala:175-215
      SlotRound key{m->p2a.slot, m->p2a.round};
      if (plStates.count(key)) continue;
      for (int r = 0; r < R; ++r) {
        auto a = std::make_unique<Msg>();
        a->kind = 1;
        a->acceptor = r;
        a->p2a = m->p2a;
        transport.push_back(std::move(a));
      }
      PlState st;
      st.pending = std::make_unique<Pending>();
      st.pending->phase2a = m->p2a;
      plStates.emplace(key, std::move(st));
    } else if (m->kind == 1) {  // Acceptor.handlePhase2a, Acceptor.scala:184-220
      Acceptor& a = acceptors[(size_t)m->acceptor];
      if (m->p2a.round < a.round) continue;  // Nack (never happens on the steady stream)
      a.round = m->p2a.round;
      a.states[m->p2a.slot] = State{a.round, m->p2a.value};
      if (m->p2a.slot > a.maxVotedSlot) a.maxVotedSlot = m->p2a.slot;
      auto b = std::make_unique<Msg>();
      b->kind = 2;
      b->p2b = Phase2b{0, m->acceptor, m->p2a.slot, a.round};
      transport.push_back(std::move(b));
    } else {  // ProxyLeader.handlePhase2b, ProxyLeader.scala:217-258
      auto it = plStates.find(SlotRound{m->p2b.slot, m->p2b.round});
      if (it == plStates.end()) return -1;  // logger.fatal
      if (it->second.done) continue;
      Pending& p = *it->second.pending;
      p.phase2bs[{m->p2b.groupIndex, m->p2b.acceptorIndex}] = m->p2b;
      if ((int)p.phase2bs.size() < f + 1) continue;
      ++chosen;
      sum += p.phase2a.value;
      it->second.done = true;
      it->second.pending.reset();
    }
  }
  if (checksum) *checksum = sum;
  return chosen;
}
