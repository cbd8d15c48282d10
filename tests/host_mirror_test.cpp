// host_mirror_test.cpp -- the reference's own unit tests for this path, written against the C++ host
// mirror (frankenpaxos_amd/host/fpx.hpp) so that they read like the Scala originals:
//   shared/src/test/scala/quorums/{GridTest,SimpleMajorityTest,UnanimousWrites}.scala
//   shared/src/test/scala/roundsystem/RoundSystemTest.scala:13-62
// plus BASELINE.json configs[0]: MultiPaxos f = 1, 1000 commands through proxy leader + acceptors.
// Needs a GPU (every predicate / handler runs in libfpx).  Built and run by tests/test_host_mirror.py.
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

int main() {
  gridTest();
  simpleMajorityTest();
  unanimousWritesTest();
  roundSystemTest();
  multiPaxos1kCommands(false);
  multiPaxos1kCommands(true);
  if (failures) {
    std::printf("%d failure(s)\n", failures);
    return 1;
  }
  std::printf("host mirror: all tests passed\n");
  return 0;
}
