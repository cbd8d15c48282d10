// host_mirror_test.cpp -- the reference's own unit tests for this path, written against the C++ host
// mirror (frankenpaxos_amd/host/fpx.hpp) so that they read like the Scala originals:
//   shared/src/test/scala/quorums/{GridTest,SimpleMajorityTest,UnanimousWrites}.scala
//   shared/src/test/scala/roundsystem/RoundSystemTest.scala:13-62
// plus BASELINE.json configs[0]: MultiPaxos f = 1, 1000 commands through proxy leader + acceptors, a leader
// change (Phase1a -> Phase1b safe values -> replica log), Mencius noop ranges and two EPaxos pre-accept ticks.
// Needs a GPU (every predicate / handler runs in libfpx) -- except the dependency-graph tests
// (shared/src/test/scala/depgraph/{DependencyGraphTest,ZigzagTarjanDependencyGraphTest}.scala), which are host code
// and run alone with `host_mirror_test --host-only`.  Built and run by tests/test_host_mirror.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "../frankenpaxos_amd/host/fpx.hpp"

using namespace frankenpaxos;
using S = std::set<int>;

static int failures = 0;
#define SHOULD_BE(expr, want)                                                        \
  do {                                                                               \
    if ((expr) != (want)) {                                                          \
      std::printf("FAIL %s:%d  %s shouldBe %s\n", __FILE__, __LINE__, #expr, #want); \
      ++failures;                                                                    \
    }                                                                                \
  } while (0)
#define SHOULD_THROW(expr)                                                    \
  do {                                                                        \
    bool thrown = false;                                                      \
    try { (void)(expr); } catch (const std::invalid_argument&) { thrown = true; } \
    if (!thrown) {                                                            \
      std::printf("FAIL %s:%d  %s should throw\n", __FILE__, __LINE__, #expr); \
      ++failures;                                                             \
    }                                                                         \
  } while (0)

static void gridTest() {  // quorums/GridTest.scala
  quorums::Grid qs({{1, 2, 3}, {4, 5, 6}});
  SHOULD_BE(qs.isReadQuorum(S{}), false);
  for (int i = 1; i <= 6; ++i) SHOULD_BE(qs.isReadQuorum(S{i}), false);
  for (int i = 1; i <= 6; ++i)
    for (int j = 1; j <= 6; ++j) SHOULD_BE(qs.isReadQuorum(S{i, j}), false);
  SHOULD_BE(qs.isReadQuorum(S{1, 2, 4}), false);
  SHOULD_BE(qs.isReadQuorum(S{4, 5, 3}), false);
  SHOULD_BE(qs.isReadQuorum(S{1, 2, 3}), true);
  SHOULD_BE(qs.isReadQuorum(S{4, 5, 6}), true);
  SHOULD_BE(qs.isReadQuorum(S{1, 2, 3, 4}), true);
  SHOULD_BE(qs.isReadQuorum(S{1, 2, 3, 4, 5}), true);
  SHOULD_BE(qs.isReadQuorum(S{1, 2, 3, 4, 5, 6}), true);

  SHOULD_BE(qs.isWriteQuorum(S{}), false);
  for (int i = 1; i <= 6; ++i) SHOULD_BE(qs.isWriteQuorum(S{i}), false);
  SHOULD_BE(qs.isWriteQuorum(S{1, 2}), false);
  SHOULD_BE(qs.isWriteQuorum(S{1, 2, 3}), false);
  SHOULD_BE(qs.isWriteQuorum(S{4, 5}), false);
  SHOULD_BE(qs.isWriteQuorum(S{4, 5, 6}), false);
  SHOULD_BE(qs.isWriteQuorum(S{1, 4}), true);
  SHOULD_BE(qs.isWriteQuorum(S{2, 4}), true);
  SHOULD_BE(qs.isWriteQuorum(S{2, 5}), true);
  SHOULD_BE(qs.isWriteQuorum(S{1, 2, 4}), true);
  SHOULD_BE(qs.isWriteQuorum(S{1, 2, 4, 5}), true);
  SHOULD_BE(qs.isWriteQuorum(S{1, 2, 3, 4, 5}), true);
  SHOULD_BE(qs.isWriteQuorum(S{1, 2, 3, 4, 5, 6}), true);

  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{9001, 1, 2, 4}), false);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{9001, 4, 5, 3}), false);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{9001, 1, 2, 3}), true);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{9001, 4, 5, 6}), true);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{9001, 1, 2, 3, 4, 5, 6}), true);

  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 1, 2}), false);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 1, 2, 3}), false);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 4, 5}), false);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 4, 5, 6}), false);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 1, 4}), true);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 2, 4}), true);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 2, 5}), true);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 1, 2, 4, 5}), true);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{9001, 1, 2, 3, 4, 5, 6}), true);
  SHOULD_THROW(qs.isWriteQuorum(S{9001, 1, 4}));  // Grid.scala:44-47 require
  SHOULD_THROW(quorums::Grid({{1, 2}, {3}}));     // Grid.scala:14-17
}

static void simpleMajorityTest() {  // quorums/SimpleMajorityTest.scala
  quorums::SimpleMajority qs(S{0, 1, 2, 3, 4});
  SHOULD_BE(qs.isReadQuorum(S{}), false);
  SHOULD_BE(qs.isReadQuorum(S{0}), false);
  SHOULD_BE(qs.isReadQuorum(S{0, 1}), false);
  SHOULD_BE(qs.isReadQuorum(S{0, 1, 2}), true);
  SHOULD_BE(qs.isReadQuorum(S{0, 1, 2, 3}), true);
  SHOULD_BE(qs.isReadQuorum(S{0, 1, 2, 3, 4}), true);
  SHOULD_BE(qs.isWriteQuorum(S{0, 1}), false);
  SHOULD_BE(qs.isWriteQuorum(S{0, 1, 2}), true);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{5}), false);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{0, 1, 5}), false);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{0, 1, 2, 5}), true);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{0, 1, 5}), false);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{0, 1, 2, 3, 4, 5}), true);
  SHOULD_THROW(qs.isWriteQuorum(S{0, 1, 5}));  // SimpleMajority.scala:42-45
  SHOULD_THROW(quorums::SimpleMajority(S{}));  // :23-26
}

static void unanimousWritesTest() {  // quorums/UnanimousWrites.scala (test)
  quorums::UnanimousWrites qs(S{0, 1, 2, 3, 4});
  SHOULD_BE(qs.isReadQuorum(S{}), false);
  SHOULD_BE(qs.isReadQuorum(S{3}), true);
  SHOULD_BE(qs.isReadQuorum(S{0, 1, 2, 3, 4}), true);
  SHOULD_BE(qs.isWriteQuorum(S{0, 1, 2, 3}), false);
  SHOULD_BE(qs.isWriteQuorum(S{0, 1, 2, 3, 4}), true);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{5}), false);
  SHOULD_BE(qs.isSuperSetOfReadQuorum(S{4, 5}), true);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{0, 1, 2, 3, 5}), false);
  SHOULD_BE(qs.isSuperSetOfWriteQuorum(S{0, 1, 2, 3, 4, 5}), true);
}

static void roundSystemTest() {  // roundsystem/RoundSystemTest.scala:13-62
  roundsystem::ClassicRoundRobin rs(3);
  SHOULD_BE(rs.numLeaders(), 3);
  const int leaders[9] = {0, 1, 2, 0, 1, 2, 0, 1, 2};
  for (int r = 0; r < 9; ++r) SHOULD_BE(rs.leader(r), leaders[r]);
  const int want[3][8] = {{0, 3, 3, 3, 6, 6, 6, 9}, {1, 1, 4, 4, 4, 7, 7, 7}, {2, 2, 2, 5, 5, 5, 8, 8}};
  for (int l = 0; l < 3; ++l)
    for (int r = -1; r <= 6; ++r) SHOULD_BE(rs.nextClassicRound(l, r), want[l][r + 1]);
}

// BASELINE.json configs[0]: MultiPaxos f = 1, 1 active leader of 2, 3 acceptor groups x 3 acceptors
// (the reference's test harness builds numAcceptorGroups + 1 groups: T/multipaxos/MultiPaxos.scala:28,49-52),
// 1000 commands, thrifty random f+1 quorums like ProxyLeader.scala:190-191, Phase2b's delivered in a
// shuffled order with duplicates.
static void multiPaxos1kCommands(bool flexible) {
  multipaxos::Config config;
  config.f = 1;
  config.numLeaders = 2;
  config.numAcceptorGroups = flexible ? 2 : 3;
  config.acceptorsPerGroup = flexible ? 2 : 3;
  config.flexible = flexible;
  config.numSlots = 1024;
  multipaxos::Phase2Engine engine(config);
  std::mt19937 rng(7);
  const int N = 1000;
  std::vector<multipaxos::Phase2a> phase2as;
  std::vector<std::vector<std::pair<int, int>>> targets;
  for (int slot = 0; slot < N; ++slot) {
    phase2as.push_back({slot, 0, 100000 + slot});
    std::vector<std::pair<int, int>> quorum;
    if (!flexible) {  // a random f+1 of the slot's group
      const int g = slot % config.numAcceptorGroups;
      int a = (int)(rng() % 3), b = (a + 1 + (int)(rng() % 2)) % 3;
      quorum = {{g, a}, {g, b}};
    } else {  // a random column (Grid.scala:30-33)
      const int col = (int)(rng() % config.acceptorsPerGroup);
      for (int row = 0; row < config.numAcceptorGroups; ++row) quorum.push_back({row, col});
    }
    targets.push_back(quorum);
  }
  std::vector<bool> fresh = engine.proxyLeaderHandlePhase2a(phase2as);
  for (bool b : fresh) SHOULD_BE(b, true);
  fresh = engine.proxyLeaderHandlePhase2a({phase2as[5]});
  SHOULD_BE((bool)fresh[0], false);  // already known: ignored (ProxyLeader.scala:177-184)
  std::vector<multipaxos::Phase2b> phase2bs;
  std::vector<multipaxos::Nack> nacks;
  engine.acceptorsHandlePhase2a(phase2as, targets, &phase2bs, &nacks);
  SHOULD_BE(nacks.size(), (size_t)0);
  SHOULD_BE(phase2bs.size(), (size_t)(2 * N));
  // half of the votes first (no quorum anywhere in non-flexible mode), then everything, shuffled
  std::vector<multipaxos::Phase2b> firstHalf;
  for (size_t i = 0; i < phase2bs.size(); i += 2) firstHalf.push_back(phase2bs[i]);
  std::vector<multipaxos::Chosen> chosen = engine.proxyLeaderHandlePhase2b(firstHalf);
  SHOULD_BE(chosen.size(), (size_t)0);
  std::shuffle(phase2bs.begin(), phase2bs.end(), rng);
  chosen = engine.proxyLeaderHandlePhase2b(phase2bs);
  SHOULD_BE(chosen.size(), (size_t)N);
  std::set<int> slots;
  for (auto& c : chosen) {
    slots.insert(c.slot);
    SHOULD_BE(c.value, 100000 + c.slot);
  }
  SHOULD_BE(slots.size(), (size_t)N);
  SHOULD_BE(engine.proxyLeaderHandlePhase2b(phase2bs).size(), (size_t)0);  // after Done: ignored
  // a Phase2b in a round the proxy leader never proposed is fatal (ProxyLeader.scala:220-225)
  bool fatal = false;
  try {
    engine.proxyLeaderHandlePhase2b({{0, 0, 3, 9}});
  } catch (const std::logic_error&) {
    fatal = true;
  }
  SHOULD_BE(fatal, true);
  // a stale leader (round 0) after the acceptors promised round 1: every acceptor Nacks
  std::vector<multipaxos::Phase2a> r1;
  for (int slot = 0; slot < 9; ++slot) r1.push_back({slot, 1, 100000 + slot});
  SHOULD_BE(engine.handlePhase2(r1).size(), (size_t)9);  // fused tick in round 1: all chosen again
  phase2bs.clear();
  engine.acceptorsHandlePhase2a({{1000, 0, 5}, {1001, 0, 5}, {1002, 0, 5}}, {}, &phase2bs, &nacks);
  SHOULD_BE(phase2bs.size(), (size_t)0);
  SHOULD_BE(nacks.size(), (size_t)3);
  for (auto& nk : nacks) {
    SHOULD_BE(nk.round, 1);
    SHOULD_BE(nk.leaderIndex, 0);
  }
}

using V = std::vector<int32_t>;

// multipaxos leader change: Acceptor.handlePhase1a -> Leader.handlePhase1b (safeValue) -> re-proposal ->
// Replica.handleChosen.  Values worked out by hand from Acceptor.scala:148-220, Leader.scala:306-329, 543-566.
static void multiPaxosRecovery() {
  multipaxos::Config config;
  config.f = 1, config.numLeaders = 2, config.numAcceptorGroups = 2, config.acceptorsPerGroup = 3;
  config.numSlots = 64;
  multipaxos::Phase2Engine engine(config);
  // round 0 (leader 0): slots 0..5; slot 2 reaches only acceptor 0 of its group, slot 5 reaches nobody useful
  std::vector<multipaxos::Phase2a> r0;
  std::vector<std::vector<std::pair<int, int>>> targets;
  for (int slot = 0; slot < 5; ++slot) {
    r0.push_back({slot, 0, 500 + slot});
    if (slot == 2) targets.push_back({{0, 0}}); else targets.push_back({});
  }
  std::vector<multipaxos::Chosen> chosen = engine.handlePhase2(r0, targets);
  SHOULD_BE(chosen.size(), (size_t)4);  // slot 2 has one vote of the f + 1 = 2 it needs
  SHOULD_BE(engine.replicaHandleChosen(chosen), 2);  // executes 0, 1; hole at 2
  SHOULD_BE(engine.numChosen(), 4);
  // leader 1 takes over in round 1: all acceptors promise (1 > 0)
  SHOULD_BE(engine.acceptorsHandlePhase1a(0, 1, 2), (std::vector<int>{0, 1, 2}));
  SHOULD_BE(engine.acceptorsHandlePhase1a(1, 1, 2, {1, 2}), (std::vector<int>{1, 2}));
  // a stale Phase1a (round 0 < 1) is Nacked by everyone it reaches (Acceptor.scala:155-163)
  SHOULD_BE(engine.acceptorsHandlePhase1a(0, 0, 2).size(), (size_t)0);
  // Phase1b's of acceptors {1, 2} of group 0 and {1, 2} of group 1, from chosenWatermark 2
  std::vector<multipaxos::Phase2Engine::SafeValue> safe = engine.leaderHandlePhase1bs(2, {{1, 2}, {1, 2}});
  SHOULD_BE(safe.size(), (size_t)3);  // maxSlot = 4
  // slot 2: its only vote is at acceptor 0, outside the quorum -> nothing voted -> Noop is safe
  SHOULD_BE(safe[0].slot, 2); SHOULD_BE(safe[0].voteRound, -1); SHOULD_BE(safe[0].value, FPX_NOOP);
  SHOULD_BE(safe[1].slot, 3); SHOULD_BE(safe[1].voteRound, 0); SHOULD_BE(safe[1].value, 503);
  SHOULD_BE(safe[2].slot, 4); SHOULD_BE(safe[2].voteRound, 0); SHOULD_BE(safe[2].value, 504);
  // with acceptor 0 in the quorum the vote for slot 2 is seen
  safe = engine.leaderHandlePhase1bs(2, {{0, 1}, {1, 2}});
  SHOULD_BE(safe[0].voteRound, 0); SHOULD_BE(safe[0].value, 502);
  // re-propose the safe values in round 1; everything is chosen, the replica's log fills up
  std::vector<multipaxos::Phase2a> r1;
  for (auto& sv : safe) r1.push_back({sv.slot, 1, sv.value});
  chosen = engine.handlePhase2(r1);
  SHOULD_BE(chosen.size(), (size_t)3);
  SHOULD_BE(engine.replicaHandleChosen(chosen), 5);
  SHOULD_BE(engine.numChosen(), 5);  // 3 and 4 were chosen before: redundantly chosen, not counted again
  // the read path (Acceptor.scala:222-254): maxVotedSlot of every acceptor.  Slots 0, 2, 4 belong to group 0 (slot % 2), 1 and 3
  // to group 1; round 1 re-proposed 2, 3, 4 everywhere
  for (int a = 0; a < 3; ++a) SHOULD_BE(engine.acceptorMaxVotedSlot(0, a), 4);
  for (int a = 0; a < 3; ++a) SHOULD_BE(engine.acceptorMaxVotedSlot(1, a), 3);
  SHOULD_BE(engine.acceptorMaxVotedSlot(0, 1, 0, 3), 2);   // among the slots 0 .. 2 only
  SHOULD_BE(engine.acceptorMaxVotedSlot(1, 2, 4, 60), -1);  // nothing of group 1 from slot 4 on
}

// mencius noop ranges, mencius/Acceptor.scala:237-291 and mencius/ProxyLeader.scala:255-303, 355-411
static void menciusNoopRange() {
  mencius::Config config;
  config.f = 1, config.numLeaderGroups = 3, config.numAcceptorGroups = 2, config.numSlots = 256;
  mencius::NoopRangeEngine engine(config);
  // acceptor 2 of (leader group 1, acceptor group 0) was promised round 5 by someone else
  const uint64_t only2[4] = {1ull << 2, 0, 0, 0};
  SHOULD_BE(fpx_acceptor_phase1a(engine.context(), 1 * 2 + 0, 5, 0, only2, nullptr, nullptr), FPX_OK);
  // leader group 1 owns slots 1, 4, 7, ...; it skips [4, 14) in round 2
  const mencius::Phase2aNoopRange range{4, 14, 2};
  bool fatal = false;
  try {
    engine.proxyLeaderHandlePhase2bNoopRange({{0, 0, 4, 14, 2}});
  } catch (const std::logic_error&) {
    fatal = true;
  }
  SHOULD_BE(fatal, true);  // never opened: logger.fatal (ProxyLeader.scala:360-366)
  SHOULD_BE(engine.proxyLeaderHandlePhase2aNoopRange(range), true);
  SHOULD_BE(engine.proxyLeaderHandlePhase2aNoopRange(range), false);  // known: ignored
  std::vector<mencius::Phase2bNoopRange> phase2bs;
  std::vector<mencius::Nack> nacks;
  engine.acceptorsHandlePhase2aNoopRange(range, &phase2bs, &nacks);
  SHOULD_BE(phase2bs.size(), (size_t)5);  // acceptor group 0: {0, 1} (2 Nacks), acceptor group 1: {0, 1, 2}
  SHOULD_BE(nacks.size(), (size_t)1);
  SHOULD_BE(nacks[0].round, 5);
  for (auto& m : phase2bs) SHOULD_BE(m.acceptorGroupIndex == 0 && m.acceptorIndex == 2, false);
  // f + 1 = 2 votes from EVERY acceptor group
  SHOULD_BE(engine.proxyLeaderHandlePhase2bNoopRange({{0, 0, 4, 14, 2}, {0, 1, 4, 14, 2}}).has_value(), false);
  SHOULD_BE(engine.proxyLeaderHandlePhase2bNoopRange({{1, 2, 4, 14, 2}}).has_value(), false);
  std::optional<mencius::ChosenNoopRange> chosen = engine.proxyLeaderHandlePhase2bNoopRange({{1, 0, 4, 14, 2}});
  SHOULD_BE(chosen.has_value(), true);
  SHOULD_BE(chosen->slotStartInclusive, 4);
  SHOULD_BE(chosen->slotEndExclusive, 14);
  SHOULD_BE(engine.proxyLeaderHandlePhase2bNoopRange(phase2bs).has_value(), false);  // Done: ignored
  // the votes are in the acceptors' logs: (round 2, Noop) in the slots of [4, 14) that leader group 1 owns
  std::vector<int32_t> voteRound(256), voteValue(256);
  SHOULD_BE(fpx_read_acceptor(engine.context(), 1 * 2 + 1, 0, nullptr, nullptr, voteRound.data(), voteValue.data(), nullptr),
            FPX_OK);
  SHOULD_BE(voteRound[4], 2); SHOULD_BE(voteValue[4], FPX_NOOP);    // slot 4 -> acceptor group (4 / 3) % 2 = 1
  SHOULD_BE(voteRound[10], 2); SHOULD_BE(voteRound[7], -1);         // slot 7 belongs to acceptor group 0
  SHOULD_BE(voteRound[16], -1);                                     // outside the range
  // the replica: slots 4, 7, 10, 13 become Noop; nothing executes (hole at 0)
  SHOULD_BE(engine.replicaHandleChosenNoopRange(*chosen), 0);
  SHOULD_BE(engine.numChosen(), 4);
  SHOULD_BE(engine.replicaHandleChosenNoopRange({4, 14}), 0);  // 4 is already chosen: returns at once
  SHOULD_BE(engine.numChosen(), 4);
  SHOULD_BE(engine.replicaHandleChosenNoopRange({0, 4}), 1);   // slots 0, 3: executes 0, hole at 1
  SHOULD_BE(engine.numChosen(), 6);
}

// epaxos pre-accept, the two hand-worked ticks of tests/test_epaxos.py (Replica.scala:569-600, 633-729,
// 1159-1419; KeyValueStore.scala:259-302)
static void epaxosPreAccept() {
  {
    epaxos::PreAcceptEngine engine(1, 4);  // n = 3: the leader asks one other replica, always fast
    std::vector<epaxos::Proposal> tick = {
        {{0, 0}, {1, true}, {1}},   // A = set k1 by replica 0, asks replica 1
        {{1, 0}, {1, false}, {2}},  // B = get k1 by replica 1, asks replica 2
        {{2, 0}, {1, true}, {0}},   // C = set k1 by replica 2, asks replica 0
        {{0, 1}, {2, false}, {2}},  // D = get k2 by replica 0, asks replica 2
    };
    // replica 0 sees A, C, D; replica 1 sees B, A; replica 2 sees C, B, D
    std::vector<epaxos::Decision> d = engine.handleTick(tick, {{0, 2, 3, 1}, {1, 0, 2, 3}, {2, 1, 3, 0}});
    SHOULD_BE(d[0].preAcceptDependencies, (V{0, 0, 0})); SHOULD_BE(d[0].dependencies, (V{0, 1, 0}));
    SHOULD_BE(d[1].preAcceptDependencies, (V{0, 0, 0})); SHOULD_BE(d[1].dependencies, (V{0, 0, 1}));
    SHOULD_BE(d[2].preAcceptDependencies, (V{0, 0, 0})); SHOULD_BE(d[2].dependencies, (V{1, 0, 0}));
    SHOULD_BE(d[3].dependencies, (V{0, 0, 0}));
    for (auto& x : d) SHOULD_BE(x.fastPath, true);
    for (int r = 0; r < 3; ++r) {
      SHOULD_BE(engine.conflictIndex(r, 1).first, (V{0, 1, 0}));
      SHOULD_BE(engine.conflictIndex(r, 1).second, (V{1, 0, 1}));
      SHOULD_BE(engine.conflictIndex(r, 2).first, (V{2, 0, 0}));
    }
    // Replica.commit -> dependencyGraph.commit -> execute (Replica.scala:859-917): A <- B <- C <- A is one component
    // (all sequence numbers 0: executed in key order), D depends on nothing
    depgraph::DependencyGraph graph(3);
    epaxos::commitToGraph(graph, tick, d, std::vector<bool>(tick.size(), true));
    auto run = graph.executeByComponent();
    using K = depgraph::Key;
    SHOULD_BE(run.first, (std::vector<std::vector<K>>{{{0, 0}, {1, 0}, {2, 0}}, {{0, 1}}}));
    SHOULD_BE(run.second, (std::set<K>{{0, 2}, {1, 1}, {2, 1}}));
    std::vector<epaxos::Proposal> tick2 = {{{1, 1}, {1, false}, {0}}};
    d = engine.handleTick(tick2);
    SHOULD_BE(d[0].fastPath, true);
    SHOULD_BE(d[0].dependencies, (V{1, 0, 1}));
    epaxos::commitToGraph(graph, tick2, d, {true});
    SHOULD_BE(graph.execute().first, (std::vector<K>{{1, 1}}));
  }
  {
    epaxos::PreAcceptEngine engine(2, 2);  // n = 5: answers that differ force the slow path
    std::vector<epaxos::Proposal> tick = {
        {{4, 0}, {0, true}, {1, 2, 3}},  // X = set k0 by replica 4
        {{0, 7}, {0, true}, {1, 2, 3}},  // Y = set k0 by replica 0
    };
    // replicas 1, 2 process X then Y; replica 3 processes Y then X
    std::vector<epaxos::Decision> d = engine.handleTick(tick, {{0, 1}, {0, 1}, {0, 1}, {1, 0}, {0, 1}});
    SHOULD_BE(d[1].fastPath, false); SHOULD_BE(d[1].dependencies, (V{0, 0, 0, 0, 1}));
    SHOULD_BE(d[0].fastPath, false); SHOULD_BE(d[0].dependencies, (V{8, 0, 0, 0, 0}));
    SHOULD_THROW(engine.handleTick({{{0, 0}, {0, true}, {1, 2}}}));     // a fast quorum has n - 2 others
    SHOULD_THROW(engine.handleTick({{{0, 0}, {0, true}, {0, 1, 2}}}));  // the leader does not ask itself
  }
}

// handlePreAccept in full, Prepare and the Accept phase on the command log: steps a) - e) of
// tests/test_epaxos.py::test_oracle_handle_preaccept_every_branch_by_hand (Replica.scala:1159-1289, 1421-1565, 1632-1757)
static void epaxosCommandLog() {
  using namespace epaxos;
  PreAcceptEngine engine(2, 4, 0, 16);  // n = 5, 4 keys, 16 instances per leader
  const Instance X{0, 0};
  // replica 1 alone knows (2, 6) on k1: a PreAccept of leader 2 delivered to it
  InstanceMessage pre26{{2, 6}, {0, 2}, {1}, 7, {1, true}, {}, 0};
  SHOULD_BE(engine.handlePreAccept({pre26})[0].ok, (std::vector<int>{1}));
  // tick: X = set k1 led by 0 pre-accepts at {1, 2, 3}: replica 1 answers [0,0,7,0,0], the others zeros -> slow path
  std::vector<Decision> d = engine.handleTick({{X, {1, true}, {1, 2, 3}}});
  SHOULD_BE(d[0].fastPath, false);
  SHOULD_BE(d[0].dependencies, (V{0, 0, 7, 0, 0}));
  SHOULD_BE((int)engine.cmdLog(1, X).kind, (int)EntryKind::PreAccepted);
  SHOULD_BE(engine.cmdLog(1, X).dependencies, (V{0, 0, 7, 0, 0}));
  SHOULD_BE((int)engine.cmdLog(4, X).kind, (int)EntryKind::None);
  // a) the leader re-sends PreAccept(X, Ballot(0,0)) to {1, 2, 4}: 1 and 2 answer again from the stored triple, 4 processes it
  InstanceMessage again{X, {0, 0}, {1, 2, 4}, 42, {1, true}, {}, 0};
  InstanceReplies a = engine.handlePreAccept({again})[0];
  SHOULD_BE(a.ok, (std::vector<int>{4}));
  SHOULD_BE(a.resent, (std::vector<int>{1, 2}));
  SHOULD_BE(a.replyDependencies[1], (V{0, 0, 7, 0, 0}));
  SHOULD_BE(a.replyDependencies[4], (V{0, 0, 0, 0, 0}));
  // b) Prepare(X, Ballot(1,2)) at 1; the old PreAccept(X, Ballot(0,0)) is now Nacked there with largestBallot (1,2)
  SHOULD_BE(engine.handlePrepare({{X, {1, 2}, {1}}})[0].ok, (std::vector<int>{1}));
  again.recipients = {1};
  InstanceReplies b = engine.handlePreAccept({again})[0];
  SHOULD_BE(b.nacks, (std::vector<int>{1}));
  SHOULD_BE(b.nackBallot == (Ballot{1, 2}), true);
  // c) PreAccept(X, Ballot(2,3), deps [0,0,0,2,0]) at {1, 2}: processed afresh, X itself taken out of its conflicts
  InstanceMessage higher{X, {2, 3}, {1, 2}, 43, {1, true}, {0, 0, 0, 2, 0}, 0};
  InstanceReplies c = engine.handlePreAccept({higher})[0];
  SHOULD_BE(c.ok, (std::vector<int>{1, 2}));
  SHOULD_BE(c.replyDependencies[1], (V{0, 0, 7, 2, 0}));
  SHOULD_BE(c.replyDependencies[2], (V{0, 0, 0, 2, 0}));
  SHOULD_BE(engine.cmdLog(1, X).ballot == (Ballot{2, 3}), true);
  // d) Accept(X, Ballot(3,4)) by 4 at 1: two AcceptOks are no quorum; PreAccept in that very ballot is then ignored
  InstanceReplies acc = engine.acceptPhase({{X, {3, 4}, {1}, 44}})[0];
  SHOULD_BE(acc.ok, (std::vector<int>{1, 4}));
  SHOULD_BE(acc.committed, false);
  InstanceMessage same{X, {3, 4}, {1, 4}, 44, {1, true}, {}, 0};
  InstanceReplies ig = engine.handlePreAccept({same})[0];
  SHOULD_BE(ig.ok.empty() && ig.resent.empty() && ig.nacks.empty() && ig.commits.empty(), true);
  // e) Accept(X, Ballot(5,2)) at {0, 1, 3} commits; any PreAccept(X) is answered with the Commit
  SHOULD_BE(engine.acceptPhase({{X, {5, 2}, {0, 1, 3}, 46}})[0].committed, true);
  InstanceMessage late{X, {0, 1}, {0, 1, 2, 3, 4}, 47, {1, true}, {}, 0};
  InstanceReplies e = engine.handlePreAccept({late})[0];
  SHOULD_BE(e.commits, (std::vector<int>{0, 1, 2, 3, 4}));
  SHOULD_BE(e.replyTripleId, (V{46, 46, 46, 46, 46}));
  SHOULD_BE((int)engine.cmdLog(3, X).kind, (int)EntryKind::Committed);
  // the proposer of an Accept that holds a CommittedEntry would have died in logger.fatal (Replica.scala:740-744)
  {
    bool fatal = false;
    try { (void)engine.acceptPhase({{X, {6, 0}, {1}, 48}}); } catch (const std::logic_error&) { fatal = true; }
    SHOULD_BE(fatal, true);
  }
  // f) recovery of another instance Y = (2, 3), pre-accepted in Ballot(0, 2) at {0, 1, 3}: replica 4 prepares it in
  //    Ballot(1, 4), holds three PrepareOk(PreAccepted) and -- as the reference evaluates handlePrepareOk -- pre-accepts
  //    the command again; as its comments intend, f = 2 identical default-ballot pre-accepts go to the Accept phase
  {
    const Instance Y{2, 3};
    InstanceMessage first{Y, {0, 2}, {0, 1, 3}, 60, {0, true}, {}, 0};
    SHOULD_BE(engine.handlePreAccept({first})[0].ok, (std::vector<int>{0, 1, 3}));
    std::vector<InstanceMessage> prep = {{Y, {1, 4}, {0, 1, 3}}};
    std::vector<InstanceReplies> oks = engine.handlePrepare(prep);
    SHOULD_BE(oks[0].ok, (std::vector<int>{0, 1, 3}));
    SHOULD_BE(oks[0].replyStatus, (V{2, 2, -1, 2, -1}));
    RecoveryDecision asWritten = engine.handlePrepareOks(prep, oks)[0];
    SHOULD_BE((int)asWritten.action, (int)RecoveryDecision::PreAcceptCommand);
    SHOULD_BE(asWritten.tripleId, 60);
    RecoveryDecision asIntended = engine.handlePrepareOks(prep, oks, true)[0];
    SHOULD_BE((int)asIntended.action, (int)RecoveryDecision::AcceptPhase);
    SHOULD_BE(asIntended.source, 0);
    oks[0].ok = {0, 1};  // only two of them in hand: no slow quorum yet
    SHOULD_BE((int)engine.handlePrepareOks(prep, oks)[0].action, (int)RecoveryDecision::Wait);
  }
  // g) a Commit from outside (Replica.handleCommit :1567-1575) for Y at {1, 2}: CommittedEntry whatever was there -- replica 1
  //    held Y PreAccepted in Ballot(1, 4) since the Prepare, replica 2 (Y's own leader) nothing --, the dependencies it carried,
  //    and a Prepare that comes later is answered with the Commit (:1746-1756)
  {
    const Instance Y{2, 3};
    engine.handleCommit(Y, 61, 0, true, {1, 2}, {1, 0, 3, 0, 2}, 0);
    SHOULD_BE((int)engine.cmdLog(1, Y).kind, (int)EntryKind::Committed);
    SHOULD_BE((int)engine.cmdLog(2, Y).kind, (int)EntryKind::Committed);
    SHOULD_BE(engine.cmdLog(2, Y).tripleId, 61);
    SHOULD_BE(engine.cmdLog(2, Y).dependencies, (V{1, 0, 3, 0, 2}));
    SHOULD_BE((int)engine.cmdLog(0, Y).kind, (int)EntryKind::PreAccepted);   // not among the recipients
    InstanceReplies p2 = engine.handlePrepare({{Y, {7, 0}, {1, 2}}})[0];
    SHOULD_BE(p2.commits, (std::vector<int>{1, 2}));
    SHOULD_BE(p2.ok.empty(), true);
    SHOULD_THROW(engine.handleCommit(Y, 61, 9, true, {1}));                  // a key outside the index
  }
  // two messages for one instance in a batch; a PreAccept that depends on itself
  SHOULD_THROW(engine.handlePrepare({{{1, 1}, {1, 0}, {2}}, {{1, 1}, {1, 0}, {3}}}));
  SHOULD_THROW(engine.handlePreAccept({{{1, 7}, {0, 1}, {0}, 1, {2, false}, {0, 8, 0, 0, 0}, 0}}));
}

// depgraph/DependencyGraphTest.scala and ZigzagTarjanDependencyGraphTest.scala through the C++ mirror (the complete
// transcription runs in tests/test_depgraph.py; these are the ones with the most structure)
static void dependencyGraphTest() {
  using namespace depgraph;
  using C = std::vector<std::vector<Key>>;
  auto ints = [](std::initializer_list<int> xs) {  // IntPrefixSet(Set(...)) with plain Int keys = column 0
    KeySet s;
    for (int x : xs) s.values.push_back({0, x});
    return s;
  };
  auto comps = [](std::initializer_list<std::initializer_list<int>> cs) {
    C out;
    for (auto& c : cs) {
      out.emplace_back();
      for (int x : c) out.back().push_back({0, x});
    }
    return out;
  };
  {  // "correctly commit a complex graph in random order" :235-256
    DependencyGraph graph(1, Kind::Tarjan);
    graph.commit({0, 6}, 1, ints({4, 5}));
    SHOULD_BE(graph.executeByComponent().first, C{});
    graph.commit({0, 4}, 0, ints({2}));
    SHOULD_BE(graph.executeByComponent().first, C{});
    graph.commit({0, 0}, 0, ints({}));
    SHOULD_BE(graph.executeByComponent().first, comps({{0}}));
    graph.commit({0, 2}, 1, ints({1}));
    SHOULD_BE(graph.executeByComponent().first, C{});
    graph.commit({0, 5}, 0, ints({3, 4, 6}));
    SHOULD_BE(graph.executeByComponent().first, C{});
    graph.commit({0, 1}, 0, ints({0, 2}));
    SHOULD_BE(graph.executeByComponent().first, comps({{1, 2}, {4}}));
    graph.commit({0, 3}, 0, ints({1, 2}));
    SHOULD_BE(graph.executeByComponent().first, comps({{3}, {5, 6}}));
    SHOULD_BE(graph.numVertices(), (int64_t)0);
  }
  {  // "correctly commit a three cycle with sequence numbers" :159-172, "report blockers chain" :488-496
    DependencyGraph graph(1, Kind::Tarjan);
    graph.commit({0, 0}, 1, ints({1}));
    graph.commit({0, 1}, 0, ints({2}));
    auto r = graph.executeByComponent();
    SHOULD_BE(r.first, C{});
    SHOULD_BE(r.second, (std::set<Key>{{0, 2}}));
    graph.commit({0, 2}, 2, ints({0}));
    SHOULD_BE(graph.executeByComponent().first, comps({{1, 0, 2}}));
  }
  {  // ZigzagTarjanDependencyGraphTest "execute a forward edge with gap correctly" :120-131
    DependencyGraph graph(3, Kind::ZigzagTarjan, 100);
    graph.commit({0, 0}, 0, KeySet{{}, {{2, 0}}});
    graph.commit({2, 0}, 1, KeySet{});
    auto r = graph.executeByComponent();
    SHOULD_BE(r.first, (C{{{2, 0}}, {{0, 0}}}));
    SHOULD_BE(r.second, (std::set<Key>{{0, 1}, {1, 0}, {2, 1}}));
    graph.commit({1, 0}, 0, KeySet{});
    graph.commit({0, 1}, 1, KeySet{});
    r = graph.executeByComponent();
    SHOULD_BE(r.first, (C{{{0, 1}}, {{1, 0}}}));
    SHOULD_BE(r.second, (std::set<Key>{{0, 2}, {1, 1}, {2, 1}}));
  }
  {  // "execute a cycle correctly" :112-118 and appendExecute (DependencyGraph.scala:160-168)
    DependencyGraph graph(3);
    graph.commit({0, 0}, 0, KeySet{{}, {{1, 0}}});
    graph.commit({1, 0}, 1, KeySet{{}, {{0, 0}}});
    std::vector<Key> executables{{9, 9}};
    std::set<Key> blockers;
    graph.appendExecute({}, executables, blockers);
    SHOULD_BE(executables, (std::vector<Key>{{9, 9}, {0, 0}, {1, 0}}));
    SHOULD_BE(blockers, (std::set<Key>{{0, 1}, {1, 1}, {2, 0}}));
    SHOULD_THROW(graph.commit({3, 0}, 0, KeySet{}));
  }
}

int main(int argc, char** argv) {
  dependencyGraphTest();
  if (argc > 1 && std::string(argv[1]) == "--host-only") {
    if (failures) {
      std::printf("%d failure(s)\n", failures);
      return 1;
    }
    std::printf("host mirror: host-only tests passed\n");
    return 0;
  }
  gridTest();
  simpleMajorityTest();
  unanimousWritesTest();
  roundSystemTest();
  multiPaxos1kCommands(false);
  multiPaxos1kCommands(true);
  multiPaxosRecovery();
  menciusNoopRange();
  epaxosPreAccept();
  epaxosCommandLog();
  if (failures) {
    std::printf("%d failure(s)\n", failures);
    return 1;
  }
  std::printf("host mirror: all tests passed\n");
  return 0;
}
