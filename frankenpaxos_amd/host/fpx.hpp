// fpx.hpp -- C++ host-side mirror of the reference's interface for the Phase-2 path, on top of the
// C ABI of include/fpx.h.  The reference is Scala (no JDK in this image), so this is the host
// language layer a user of the reference would program against: same names, argument meaning and
// error behaviour as
//
//   frankenpaxos.roundsystem.RoundSystem.ClassicRoundRobin   roundsystem/RoundSystem.scala:60-87
//   frankenpaxos.quorums.{QuorumSystem,SimpleMajority,Grid,UnanimousWrites}  quorums/*.scala
//   frankenpaxos.multipaxos.Acceptor.handlePhase2a           multipaxos/Acceptor.scala:184-220
//   frankenpaxos.multipaxos.ProxyLeader.handlePhase2a/2b     multipaxos/ProxyLeader.scala:175-258
//   frankenpaxos.multipaxos.Acceptor.handlePhase1a           multipaxos/Acceptor.scala:148-182
//   frankenpaxos.multipaxos.Replica.handleChosen             multipaxos/Replica.scala:572-590, 394-447
//   frankenpaxos.multipaxos.Leader.handlePhase1b (safeValue) multipaxos/Leader.scala:306-329, 543-566
//   frankenpaxos.mencius.{Acceptor,ProxyLeader}.handlePhase2aNoopRange / handlePhase2bNoopRange
//                                                            mencius/Acceptor.scala:237-291,
//                                                            mencius/ProxyLeader.scala:255-303, 355-411
//   frankenpaxos.epaxos.Replica pre-accept fast path         epaxos/Replica.scala:569-600, 633-729, 1159-1419
//   frankenpaxos.depgraph.{DependencyGraph, TarjanDependencyGraph, ZigzagTarjanDependencyGraph}
//                                                            depgraph/*.scala; epaxos/Replica.scala:859-917
//
// but batched: a handler takes the messages one event-loop tick delivered and returns the messages
// the reference handlers would have sent.  require(...) failures throw std::invalid_argument
// (IllegalArgumentException); logger.fatal throws std::logic_error.  Everything computes on the GPU
// through libfpx; there is no host implementation of the predicates or handlers in this file.
#pragma once

#include <cstdint>
#include <map>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fpx.h"
#include "../../include/fpx_depgraph.h"

namespace frankenpaxos {

inline void check(int32_t status, const char* what) {
  if (status == FPX_OK) return;
  const std::string msg = std::string(what) + ": " + fpx_strerror(status);
  if (status == FPX_EINVAL) throw std::invalid_argument(msg);
  if (status == FPX_EFATAL_UNKNOWN_SLOTROUND) throw std::logic_error(msg);
  throw std::runtime_error(msg);
}

namespace roundsystem {

// roundsystem/RoundSystem.scala:60-87
class ClassicRoundRobin {
 public:
  explicit ClassicRoundRobin(int n) : n_(n) {}
  int numLeaders() const { return n_; }
  int leader(int round) const { return fpx_round_leader(n_, round); }
  int nextClassicRound(int leaderIndex, int round) const { return fpx_next_classic_round(n_, leaderIndex, round); }

 private:
  int n_;
};

}  // namespace roundsystem

namespace quorums {

// quorums/QuorumSystem.scala:16-24 over Set[Int]; node ids are mapped to bit positions in
// construction order, ids outside the system map to a spare bit (a "foreign" node).
class QuorumSystem {
 public:
  virtual ~QuorumSystem() = default;
  std::set<int> nodes() const {
    std::set<int> s;
    for (auto& kv : bit_) s.insert(kv.first);
    return s;
  }
  bool isReadQuorum(const std::set<int>& xs) const { return eval(xs, true, true); }
  bool isWriteQuorum(const std::set<int>& xs) const { return eval(xs, true, false); }
  bool isSuperSetOfReadQuorum(const std::set<int>& xs) const { return eval(xs, false, true); }
  bool isSuperSetOfWriteQuorum(const std::set<int>& xs) const { return eval(xs, false, false); }
  const fpx_config& config() const { return cfg_; }

 protected:
  QuorumSystem() : cfg_{} {
    cfg_.num_slots = 1;
    cfg_.num_groups = cfg_.num_leader_groups = cfg_.num_leaders = cfg_.tally_ways = 1;
  }
  void addNode(int id) {
    if (!bit_.count(id)) {
      const int b = (int)bit_.size();
      bit_[id] = b;
    }
  }
  fpx_config cfg_;
  std::map<int, int> bit_;

 private:
  bool eval(const std::set<int>& xs, bool strict, bool read) const {
    uint64_t nodes[4] = {0, 0, 0, 0};
    for (int x : xs) {
      auto it = bit_.find(x);
      // a node that is not part of the system: strict predicates reject it (require), the superset
      // predicates ignore it.  Bit 255 is never a member (systems here have < 256 nodes).
      const int b = it == bit_.end() ? 255 : it->second;
      nodes[b >> 6] |= 1ull << (b & 63);
    }
    uint8_t out = 0;
    const int32_t st = read ? fpx_read_quorum_eval(&cfg_, 1, nodes, strict ? 1 : 0, &out)
                            : fpx_quorum_eval(&cfg_, 1, nodes, strict ? 1 : 0, &out);
    check(st, "QuorumSystem");
    return out != 0;
  }
};

// quorums/SimpleMajority.scala:19-56
class SimpleMajority : public QuorumSystem {
 public:
  explicit SimpleMajority(const std::set<int>& members) {
    if (members.empty() || members.size() > 255)
      throw std::invalid_argument("You cannot construct a SimpleMajority quorum system without any members.");
    for (int m : members) addNode(m);
    cfg_.num_replicas = (int)members.size();
    cfg_.quorum_kind = FPX_Q_SIMPLE_MAJORITY;
  }
};

// quorums/UnanimousWrites.scala:17-58
class UnanimousWrites : public QuorumSystem {
 public:
  explicit UnanimousWrites(const std::set<int>& members) {
    if (members.empty() || members.size() > 255)
      throw std::invalid_argument("You cannot construct a UnanimousWrites quorum system without any members.");
    for (int m : members) addNode(m);
    cfg_.num_replicas = (int)members.size();
    cfg_.quorum_kind = FPX_Q_UNANIMOUS;
  }
};

// quorums/Grid.scala:5-57: every row is a read quorum, one entry from every row is a write quorum
class Grid : public QuorumSystem {
 public:
  explicit Grid(const std::vector<std::vector<int>>& grid) {
    if (grid.empty()) throw std::invalid_argument("You cannot construct a Grid quorum system without any grid.");
    for (auto& row : grid)
      if (row.size() != grid[0].size()) throw std::invalid_argument("A grid quorum assumes equal sized rows.");
    for (auto& row : grid)
      for (int x : row) addNode(x);  // row-major => bit = row * cols + col
    cfg_.num_replicas = (int)(grid.size() * grid[0].size());
    if (cfg_.num_replicas > 255 || (int)bit_.size() != cfg_.num_replicas)
      throw std::invalid_argument("Grid: at most 255 distinct nodes");
    cfg_.quorum_kind = FPX_Q_GRID;
    cfg_.grid_rows = (int)grid.size();
    cfg_.grid_cols = (int)grid[0].size();
  }
};

}  // namespace quorums

namespace multipaxos {

// multipaxos/MultiPaxos.proto:273-298 (value = int32 id standing in for CommandBatchOrNoop)
struct Phase2a { int32_t slot, round, value; };
struct Phase2b { int32_t groupIndex, acceptorIndex, slot, round; };
struct Nack { int32_t leaderIndex, round; };  // destination = roundSystem.leader(phase2a.round), Acceptor.scala:197
struct Chosen { int32_t slot, value; };

// The subset of multipaxos/Config.scala:6-31 the Phase-2 path reads.
struct Config {
  int f = 1;
  int numLeaders = 2;
  int numAcceptorGroups = 1;   // non-flexible: log is round-robin partitioned over the groups
  int acceptorsPerGroup = 3;   // non-flexible: 2f+1 (Config.scala:96); flexible: columns of the grid
  bool flexible = false;       // flexible: ONE grid, rows = acceptor groups (ProxyLeader.scala:116-122)
  int numSlots = 1 << 16;      // log window held in HBM
  int tallyWays = 4;
  int device = 0;

  // Config.scala:32-147 (the require()s that concern this path)
  void checkValid() const {
    if (f < 1) throw std::invalid_argument("f must be >= 1.");
    if (numLeaders < f + 1) throw std::invalid_argument("numLeaders must be >= f + 1.");
    if (numAcceptorGroups < 1) throw std::invalid_argument("numAcceptorGroups must be >= 1.");
    if (!flexible && acceptorsPerGroup != 2 * f + 1)
      throw std::invalid_argument("acceptor clusters must have 2*f + 1 acceptors.");
    if (flexible && (acceptorsPerGroup < 1 || numAcceptorGroups * acceptorsPerGroup > 256))
      throw std::invalid_argument("grid too large");
  }

  fpx_config toFpx() const {
    fpx_config c{};
    c.num_slots = numSlots;
    c.num_leader_groups = 1;
    c.num_leaders = numLeaders;
    c.f = f;
    c.tally_ways = tallyWays;
    c.device = device;
    c.ballot_mode = FPX_BALLOT_ACCEPTOR;
    if (!flexible) {
      c.num_replicas = acceptorsPerGroup;
      c.num_groups = numAcceptorGroups;
      c.quorum_kind = FPX_Q_THRESHOLD;
    } else {
      c.num_replicas = numAcceptorGroups * acceptorsPerGroup;
      c.num_groups = 1;
      c.quorum_kind = FPX_Q_GRID;
      c.grid_rows = numAcceptorGroups;
      c.grid_cols = acceptorsPerGroup;
    }
    return c;
  }
};

// The acceptors of every group plus one proxy leader, living in the HBM of one GPU.
class Phase2Engine {
 public:
  explicit Phase2Engine(const Config& config) : config_(config), roundSystem_(config.numLeaders) {
    config.checkValid();
    fcfg_ = config.toFpx();
    check(fpx_create(&fcfg_, &ctx_), "fpx_create");
  }
  ~Phase2Engine() {
    if (ctx_) fpx_destroy(ctx_);
  }
  Phase2Engine(const Phase2Engine&) = delete;
  Phase2Engine& operator=(const Phase2Engine&) = delete;

  // ---- Acceptor.handlePhase2a (Acceptor.scala:184-220) for one tick of Phase2a's.  targets[i] is
  // the set of (groupIndex, acceptorIndex) the proxy leader sent message i to; empty => everyone.
  void acceptorsHandlePhase2a(const std::vector<Phase2a>& msgs,
                              const std::vector<std::vector<std::pair<int, int>>>& targets,
                              std::vector<Phase2b>* phase2bs, std::vector<Nack>* nacks) {
    const int n = (int)msgs.size();
    soa(msgs);
    std::vector<uint64_t> tgt = targetMasks(msgs, targets);
    std::vector<uint64_t> votes((size_t)n * 4), nk((size_t)n * 4);
    std::vector<int32_t> nround(n);
    check(fpx_acceptor_phase2a(ctx_, n, slot_.data(), round_.data(), value_.data(), tgt.empty() ? nullptr : tgt.data(),
                               votes.data(), nk.data(), nround.data()),
          "Acceptor.handlePhase2a");
    for (int i = 0; i < n; ++i) {
      for (int b = 0; b < fcfg_.num_replicas; ++b) {
        if ((votes[(size_t)i * 4 + (b >> 6)] >> (b & 63)) & 1) {
          auto ga = groupAcceptor(msgs[i].slot, b);
          phase2bs->push_back(Phase2b{ga.first, ga.second, msgs[i].slot, msgs[i].round});
        }
      }
      // every Nack of message i goes to the same leader; Leader.handleNack only reacts to the max
      if (nround[i] >= 0) nacks->push_back(Nack{roundSystem_.leader(msgs[i].round), nround[i]});
    }
  }

  // ---- ProxyLeader.handlePhase2a bookkeeping (ProxyLeader.scala:175-215): returns, per message,
  // whether it is new (true) or already known and ignored (false).
  std::vector<bool> proxyLeaderHandlePhase2a(const std::vector<Phase2a>& msgs) {
    const int n = (int)msgs.size();
    soa(msgs);
    std::vector<uint8_t> fresh(n);
    check(fpx_proxy_open(ctx_, n, slot_.data(), round_.data(), value_.data(), fresh.data()), "ProxyLeader.handlePhase2a");
    return std::vector<bool>(fresh.begin(), fresh.end());
  }

  // ---- ProxyLeader.handlePhase2b (ProxyLeader.scala:217-258): returns the Chosen messages sent to
  // the replicas.  A Phase2b for an unknown (slot, round) throws std::logic_error (logger.fatal).
  std::vector<Chosen> proxyLeaderHandlePhase2b(const std::vector<Phase2b>& msgs) {
    // group the votes of one (slot, round) into one bitmap row, preserving first-seen order
    std::map<std::pair<int, int>, int> row;
    std::vector<int32_t> slot, round;
    std::vector<uint64_t> bits;
    for (auto& m : msgs) {
      auto key = std::make_pair(m.slot, m.round);
      auto it = row.find(key);
      int r;
      if (it == row.end()) {
        r = (int)slot.size();
        row[key] = r;
        slot.push_back(m.slot);
        round.push_back(m.round);
        bits.resize(bits.size() + 4, 0);
      } else {
        r = it->second;
      }
      const int b = config_.flexible ? m.groupIndex * config_.acceptorsPerGroup + m.acceptorIndex : m.acceptorIndex;
      bits[(size_t)r * 4 + (b >> 6)] |= 1ull << (b & 63);
    }
    const int n = (int)slot.size();
    std::vector<uint8_t> ch(n);
    std::vector<int32_t> cr(n), cv(n);
    check(fpx_proxy_phase2b(ctx_, n, slot.data(), round.data(), bits.data(), ch.data(), cr.data(), cv.data()),
          "ProxyLeader.handlePhase2b");
    std::vector<Chosen> out;
    for (int i = 0; i < n; ++i)
      if (ch[i]) out.push_back(Chosen{slot[i], cv[i]});
    return out;
  }

  // ---- the fused tick: ProxyLeader.handlePhase2a -> Acceptor.handlePhase2a -> ProxyLeader.handlePhase2b
  std::vector<Chosen> handlePhase2(const std::vector<Phase2a>& msgs,
                                   const std::vector<std::vector<std::pair<int, int>>>& targets = {}) {
    const int n = (int)msgs.size();
    soa(msgs);
    std::vector<uint64_t> tgt = targetMasks(msgs, targets);
    std::vector<uint8_t> ch(n);
    std::vector<int32_t> cr(n), cv(n);
    check(fpx_phase2_fused(ctx_, n, slot_.data(), round_.data(), value_.data(), tgt.empty() ? nullptr : tgt.data(),
                           ch.data(), cr.data(), cv.data(), nullptr),
          "handlePhase2");
    std::vector<Chosen> out;
    for (int i = 0; i < n; ++i)
      if (ch[i]) out.push_back(Chosen{msgs[i].slot, cv[i]});
    return out;
  }

  // ---- Acceptor.handlePhase1a (Acceptor.scala:148-182) delivered to the acceptors of one group (empty
  // `acceptors` => all of them).  Returns the acceptor indices that answered Phase1b (their vote info
  // stays in HBM; leaderHandlePhase1bs reads it); the others Nack with their round.
  std::vector<int> acceptorsHandlePhase1a(int groupIndex, int round, int chosenWatermark,
                                          const std::vector<int>& acceptors = {}) {
    uint64_t tgt[4] = {0, 0, 0, 0}, promised[4], nack[4];
    for (int a : acceptors) {
      if (a < 0 || a >= fcfg_.num_replicas) throw std::invalid_argument("acceptor index out of range");
      tgt[a >> 6] |= 1ull << (a & 63);
    }
    check(fpx_acceptor_phase1a(ctx_, groupIndex, round, chosenWatermark, acceptors.empty() ? nullptr : tgt, promised,
                               nack),
          "Acceptor.handlePhase1a");
    std::vector<int> out;
    for (int b = 0; b < fcfg_.num_replicas; ++b)
      if ((promised[b >> 6] >> (b & 63)) & 1) out.push_back(b);
    return out;
  }

  // ---- Acceptor.handleMaxSlotRequest / handleBatchMaxSlotRequest (Acceptor.scala:222-254): what the reply's `slot` is --
  // the acceptor's maxVotedSlot, over the whole log or over the slots [firstSlot, firstSlot + count) of it
  int acceptorMaxVotedSlot(int groupIndex, int acceptorIndex, int firstSlot = 0, int count = -1) {
    int32_t slot = -1;
    check(fpx_acceptor_max_voted_in(ctx_, groupIndex, acceptorIndex, firstSlot, count < 0 ? fcfg_.num_slots - firstSlot : count, &slot),
          "Acceptor.handleMaxSlotRequest");
    return slot;
  }

  // ---- Leader.handlePhase1b once a read quorum of Phase1b's is in (Leader.scala:543-566): for every
  // slot in [chosenWatermark, maxSlot] the safe value (Leader.scala:306-329) to re-propose.
  // quorum[g] = acceptor indices of group g whose Phase1b the leader holds.
  struct SafeValue { int32_t slot, voteRound, value; };
  std::vector<SafeValue> leaderHandlePhase1bs(int chosenWatermark, const std::vector<std::vector<int>>& quorum) {
    if ((int)quorum.size() != fcfg_.num_groups) throw std::invalid_argument("one Phase1b set per acceptor group");
    std::vector<uint64_t> masks((size_t)fcfg_.num_groups * 4, 0);
    for (int g = 0; g < fcfg_.num_groups; ++g)
      for (int a : quorum[g]) masks[(size_t)g * 4 + (a >> 6)] |= 1ull << (a & 63);
    int32_t maxSlot = -1;
    check(fpx_leader_phase1b_scan(ctx_, chosenWatermark, masks.data(), 0, &maxSlot, nullptr, nullptr),
          "Leader.handlePhase1b");
    const int cap = maxSlot >= chosenWatermark ? maxSlot - chosenWatermark + 1 : 0;
    std::vector<int32_t> sr(cap), sv(cap);
    if (cap > 0)
      check(fpx_leader_phase1b_scan(ctx_, chosenWatermark, masks.data(), cap, &maxSlot, sr.data(), sv.data()),
            "Leader.handlePhase1b");
    std::vector<SafeValue> out;
    for (int k = 0; k < cap; ++k) out.push_back(SafeValue{chosenWatermark + k, sr[k], sv[k]});
    return out;
  }

  // ---- Replica.handleChosen + executeLog (Replica.scala:572-590, 394-447): returns executedWatermark
  int replicaHandleChosen(const std::vector<Chosen>& msgs) {
    std::vector<int32_t> slot(msgs.size()), value(msgs.size());
    for (size_t i = 0; i < msgs.size(); ++i) slot[i] = msgs[i].slot, value[i] = msgs[i].value;
    int32_t wm = 0, nc = 0;
    check(fpx_replica_chosen(ctx_, (int)msgs.size(), slot.data(), value.data(), nullptr, &wm, &nc),
          "Replica.handleChosen");
    numChosen_ = nc;
    return wm;
  }
  int numChosen() const { return numChosen_; }

  fpx_ctx* context() { return ctx_; }

 private:
  void soa(const std::vector<Phase2a>& msgs) {
    const size_t n = msgs.size();
    slot_.resize(n), round_.resize(n), value_.resize(n);
    for (size_t i = 0; i < n; ++i) slot_[i] = msgs[i].slot, round_[i] = msgs[i].round, value_[i] = msgs[i].value;
  }
  // bit of acceptor (groupIndex, acceptorIndex) in the row of `slot`
  int bitOf(int slot, int groupIndex, int acceptorIndex) const {
    if (config_.flexible) return groupIndex * config_.acceptorsPerGroup + acceptorIndex;
    if (groupIndex != slot % config_.numAcceptorGroups)
      throw std::invalid_argument("acceptor group does not own this slot (ProxyLeader.scala:190)");
    return acceptorIndex;
  }
  std::pair<int, int> groupAcceptor(int slot, int bit) const {
    if (config_.flexible) return {bit / config_.acceptorsPerGroup, bit % config_.acceptorsPerGroup};
    return {slot % config_.numAcceptorGroups, bit};
  }
  std::vector<uint64_t> targetMasks(const std::vector<Phase2a>& msgs,
                                    const std::vector<std::vector<std::pair<int, int>>>& targets) const {
    std::vector<uint64_t> t;
    if (targets.empty()) return t;
    if (targets.size() != msgs.size()) throw std::invalid_argument("one target set per message");
    t.assign(msgs.size() * 4, 0);
    for (size_t i = 0; i < msgs.size(); ++i) {
      if (targets[i].empty()) {
        for (int b = 0; b < fcfg_.num_replicas; ++b) t[i * 4 + (b >> 6)] |= 1ull << (b & 63);
      } else {
        for (auto& ga : targets[i]) {
          const int b = bitOf(msgs[i].slot, ga.first, ga.second);
          t[i * 4 + (b >> 6)] |= 1ull << (b & 63);
        }
      }
    }
    return t;
  }

  Config config_;
  roundsystem::ClassicRoundRobin roundSystem_;
  fpx_config fcfg_;
  fpx_ctx* ctx_ = nullptr;
  std::vector<int32_t> slot_, round_, value_;
  int numChosen_ = 0;
};

}  // namespace multipaxos

namespace mencius {

// mencius/Mencius.proto:160-203
struct Phase2aNoopRange { int32_t slotStartInclusive, slotEndExclusive, round; };
struct Phase2bNoopRange { int32_t acceptorGroupIndex, acceptorIndex, slotStartInclusive, slotEndExclusive, round; };
struct ChosenNoopRange { int32_t slotStartInclusive, slotEndExclusive; };
struct Nack { int32_t round; };

// The subset of mencius/Config.scala the noop-range path reads: numLeaderGroups leader groups own the
// slots round-robin (slot % numLeaderGroups); each has numAcceptorGroups groups of 2f+1 acceptors.
struct Config {
  int f = 1;
  int numLeaderGroups = 1;
  int numLeadersPerGroup = 2;
  int numAcceptorGroups = 1;
  int numSlots = 1 << 16;
  int tallyWays = 8;
  int device = 0;

  void checkValid() const {
    if (f < 1) throw std::invalid_argument("f must be >= 1.");
    if (numLeaderGroups < 1) throw std::invalid_argument("numLeaderGroups must be >= 1.");
    if (numLeadersPerGroup < f + 1) throw std::invalid_argument("every leader group needs >= f + 1 leaders.");
    if (numAcceptorGroups < 1) throw std::invalid_argument("numAcceptorGroups must be >= 1.");
  }
  fpx_config toFpx() const {
    fpx_config c{};
    c.num_slots = numSlots;
    c.num_replicas = 2 * f + 1;
    c.num_groups = numAcceptorGroups;
    c.num_leader_groups = numLeaderGroups;
    c.num_leaders = numLeadersPerGroup;
    c.f = f;
    c.quorum_kind = FPX_Q_THRESHOLD;
    c.ballot_mode = FPX_BALLOT_ACCEPTOR;
    c.tally_ways = tallyWays;
    c.device = device;
    return c;
  }
};

// The acceptors of every leader group plus one proxy leader, for the noop-range messages a Mencius
// leader sends when it skips its slots (mencius/Leader.scala sends Phase2aNoopRange to a proxy leader).
class NoopRangeEngine {
 public:
  explicit NoopRangeEngine(const Config& config) : config_(config) {
    config.checkValid();
    fcfg_ = config.toFpx();
    check(fpx_create(&fcfg_, &ctx_), "fpx_create");
  }
  ~NoopRangeEngine() {
    if (ctx_) fpx_destroy(ctx_);
  }
  NoopRangeEngine(const NoopRangeEngine&) = delete;
  NoopRangeEngine& operator=(const NoopRangeEngine&) = delete;

  // mencius.ProxyLeader.handlePhase2aNoopRange (ProxyLeader.scala:255-303): true = new, relayed to every
  // acceptor group; false = a known (start, end, round), ignored
  bool proxyLeaderHandlePhase2aNoopRange(const Phase2aNoopRange& m) {
    uint8_t fresh = 0;
    check(fpx_proxy_open_noop_range(ctx_, m.slotStartInclusive, m.slotEndExclusive, m.round, &fresh),
          "ProxyLeader.handlePhase2aNoopRange");
    return fresh != 0;
  }

  // mencius.Acceptor.handlePhase2aNoopRange (Acceptor.scala:237-291) at every acceptor of the owning
  // leader group's acceptor groups
  void acceptorsHandlePhase2aNoopRange(const Phase2aNoopRange& m, std::vector<Phase2bNoopRange>* phase2bs,
                                       std::vector<Nack>* nacks) {
    const int A = fcfg_.num_groups, R = fcfg_.num_replicas;
    std::vector<uint64_t> votes((size_t)A * 4), nk((size_t)A * 4);
    int32_t nackRound = -1;
    check(fpx_acceptor_phase2a_noop_range(ctx_, m.slotStartInclusive, m.slotEndExclusive, m.round, nullptr,
                                          votes.data(), nk.data(), &nackRound),
          "Acceptor.handlePhase2aNoopRange");
    for (int g = 0; g < A; ++g)
      for (int b = 0; b < R; ++b)
        if ((votes[(size_t)g * 4 + (b >> 6)] >> (b & 63)) & 1)
          phase2bs->push_back(Phase2bNoopRange{g, b, m.slotStartInclusive, m.slotEndExclusive, m.round});
    if (nackRound >= 0) nacks->push_back(Nack{nackRound});
  }

  // mencius.ProxyLeader.handlePhase2bNoopRange (ProxyLeader.scala:355-411): all messages must carry the
  // same (start, end, round); ChosenNoopRange once every acceptor group has f + 1 votes.
  std::optional<ChosenNoopRange> proxyLeaderHandlePhase2bNoopRange(const std::vector<Phase2bNoopRange>& msgs) {
    if (msgs.empty()) return std::nullopt;
    const Phase2bNoopRange& h = msgs[0];
    std::vector<uint64_t> votes((size_t)fcfg_.num_groups * 4, 0);
    for (auto& m : msgs) {
      if (m.slotStartInclusive != h.slotStartInclusive || m.slotEndExclusive != h.slotEndExclusive || m.round != h.round)
        throw std::invalid_argument("one (start, end, round) per call");
      if (m.acceptorGroupIndex < 0 || m.acceptorGroupIndex >= fcfg_.num_groups || m.acceptorIndex < 0 ||
          m.acceptorIndex >= fcfg_.num_replicas)
        throw std::invalid_argument("acceptor out of range");
      votes[(size_t)m.acceptorGroupIndex * 4 + (m.acceptorIndex >> 6)] |= 1ull << (m.acceptorIndex & 63);
    }
    uint8_t chosen = 0;
    check(fpx_proxy_phase2b_noop_range(ctx_, h.slotStartInclusive, h.slotEndExclusive, h.round, votes.data(), &chosen),
          "ProxyLeader.handlePhase2bNoopRange");
    if (!chosen) return std::nullopt;
    return ChosenNoopRange{h.slotStartInclusive, h.slotEndExclusive};
  }

  // mencius.Replica.handleChosenNoopRange (Replica.scala:464-485) at the replica whose log lives in this
  // context: returns executedWatermark.  Like the reference it stops at the first slot of the range that
  // is already chosen -- without filling the rest and without executing the log.
  int replicaHandleChosenNoopRange(const ChosenNoopRange& m) {
    int32_t wm = 0, nc = 0;
    check(fpx_replica_chosen_noop_range(ctx_, m.slotStartInclusive, m.slotEndExclusive, &wm, &nc),
          "Replica.handleChosenNoopRange");
    numChosen_ = nc;
    return wm;
  }
  int numChosen() const { return numChosen_; }

  fpx_ctx* context() { return ctx_; }

 private:
  Config config_;
  fpx_config fcfg_;
  fpx_ctx* ctx_ = nullptr;
  int numChosen_ = 0;
};

}  // namespace mencius

namespace depgraph {

// depgraph/DependencyGraph.scala:126-192 with Key = (leaderIndex, id) (epaxos.Instance; util/VertexIdLike.scala),
// SequenceNumber = Int, KeySet = an InstancePrefixSet (per leader: a watermark + explicit ids).  Host code of libfpx
// (csrc/fpx_depgraph.cpp); include/fpx_depgraph.h says which orders the reference leaves open and what is taken.
using Key = std::pair<int32_t, int32_t>;
struct KeySet {
  std::vector<int32_t> watermarks;  // one per leader; empty = all 0
  std::vector<Key> values;
};
enum class Kind { Tarjan = FPX_DG_TARJAN, ZigzagTarjan = FPX_DG_ZIGZAG };

class DependencyGraph {
 public:
  // TarjanDependencyGraph.scala:171-183 / ZigzagTarjanDependencyGraph.scala:247-262 (options.garbageCollectEveryNCommands)
  explicit DependencyGraph(int numLeaders, Kind kind = Kind::ZigzagTarjan, int garbageCollectEveryNCommands = 1000)
      : n_(numLeaders) {
    fpx_depgraph_config c{(int32_t)kind, numLeaders, garbageCollectEveryNCommands};
    if (fpx_depgraph_create(&c, &g_) != FPX_OK) throw std::invalid_argument("DependencyGraph: bad configuration");
  }
  ~DependencyGraph() {
    if (g_) fpx_depgraph_destroy(g_);
  }
  DependencyGraph(const DependencyGraph&) = delete;
  DependencyGraph& operator=(const DependencyGraph&) = delete;
  int numLeaders() const { return n_; }

  // :140-144
  void commit(Key key, int32_t sequenceNumber, const KeySet& dependencies) {
    std::vector<int32_t> wm = dependencies.watermarks.empty() ? std::vector<int32_t>(n_, 0) : dependencies.watermarks;
    if ((int)wm.size() != n_) throw std::invalid_argument("one watermark per leader");
    std::vector<int32_t> vl, vi;
    for (const Key& k : dependencies.values) vl.push_back(k.first), vi.push_back(k.second);
    const int64_t off[2] = {0, (int64_t)vl.size()};
    check(fpx_depgraph_commit(g_, 1, &key.first, &key.second, &sequenceNumber, wm.data(), off, vl.data(), vi.data()));
  }
  // a tick's committed triples in the encoding the GPU kernels produce (fpx_depgraph_commit_epx)
  void commitEpaxos(const std::vector<int32_t>& leader, const std::vector<int32_t>& id, const std::vector<int32_t>& deps,
                    const std::vector<int32_t>& ownValuesEnd) {
    check(fpx_depgraph_commit_epx(g_, (int32_t)leader.size(), leader.data(), id.data(), nullptr, deps.data(),
                                  ownValuesEnd.empty() ? nullptr : ownValuesEnd.data(), 1, nullptr));
  }
  // :170-176
  std::pair<std::vector<std::vector<Key>>, std::set<Key>> executeByComponent(std::optional<int> numBlockers = {}) {
    int64_t ne = 0, nc = 0, nb = 0;
    check(fpx_depgraph_execute(g_, numBlockers ? *numBlockers : -1, &ne, &nc, &nb));
    std::vector<int32_t> el((size_t)ne), ei((size_t)ne), cs((size_t)nc), bl((size_t)nb), bi((size_t)nb);
    check(fpx_depgraph_read_result(g_, el.data(), ei.data(), cs.data(), bl.data(), bi.data()));
    std::pair<std::vector<std::vector<Key>>, std::set<Key>> out;
    size_t at = 0;
    for (int32_t c : cs) {
      out.first.emplace_back();
      for (int32_t k = 0; k < c; ++k, ++at) out.first.back().emplace_back(el[at], ei[at]);
    }
    for (size_t k = 0; k < bl.size(); ++k) out.second.emplace(bl[k], bi[k]);
    return out;
  }
  // :146-158
  std::pair<std::vector<Key>, std::set<Key>> execute(std::optional<int> numBlockers = {}) {
    auto r = executeByComponent(numBlockers);
    std::pair<std::vector<Key>, std::set<Key>> out;
    for (auto& c : r.first) out.first.insert(out.first.end(), c.begin(), c.end());
    out.second = std::move(r.second);
    return out;
  }
  // :160-168
  void appendExecute(std::optional<int> numBlockers, std::vector<Key>& executables, std::set<Key>& blockers) {
    auto r = execute(numBlockers);
    executables.insert(executables.end(), r.first.begin(), r.first.end());
    blockers.insert(r.second.begin(), r.second.end());
  }
  // :178-186
  void updateExecuted(const KeySet& keys) {
    std::vector<int32_t> vl, vi;
    for (const Key& k : keys.values) vl.push_back(k.first), vi.push_back(k.second);
    if (!keys.watermarks.empty() && (int)keys.watermarks.size() != n_)
      throw std::invalid_argument("one watermark per leader");
    check(fpx_depgraph_update_executed(g_, keys.watermarks.empty() ? nullptr : keys.watermarks.data(),
                                       (int32_t)vl.size(), vl.data(), vi.data()));
  }
  // :188-191
  int64_t numVertices() const { return fpx_depgraph_num_vertices(g_); }

 private:
  static void check(int32_t st) {
    if (st == FPX_EINVAL) throw std::invalid_argument("DependencyGraph: require failed");
    if (st != FPX_OK) throw std::runtime_error("DependencyGraph: libfpx status " + std::to_string(st));
  }
  int n_;
  fpx_depgraph* g_ = nullptr;
};

}  // namespace depgraph

namespace epaxos {

// epaxos/EPaxos.proto:35-41
struct Instance { int32_t replicaIndex, instanceNumber; };
// a single-key key-value-store command (statemachine/KeyValueStore.scala:225-302: GetRequest / SetRequest)
struct Command { int32_t key; bool isSet; };
// one fresh instance entering the pre-accept phase in this tick
struct Proposal {
  Instance instance;
  Command command;
  std::vector<int> fastQuorum;  // the n-2 other replicas whose PreAcceptOk's the leader decides on: with a
                                // thrifty system the only ones it sends to (Replica.scala:705-706)
  std::vector<int> recipients;  // every other replica that receives the PreAccept; empty => fastQuorum.  With
                                // the reference's default ThriftySystem.NotThrifty: all n-1 others.
};
// what the leader does with it after the PreAcceptOk's are in
struct Decision {
  bool fastPath;                      // committed on the fast path (Replica.scala:1399-1405)
  std::vector<int32_t> dependencies;  // per leader replica: the dependency watermark (committed or Accept-phase)
  std::vector<int32_t> preAcceptDependencies;  // what the PreAccept carried (the leader's own conflicts)
  // IntPrefixSet `values` of the instance's own-leader column after dependencies.subtractOne(instance)
  // (Replica.scala:582): the run instanceNumber + 1 .. end - 1; 0 = none (always, on FIFO channels)
  int32_t ownValuesEnd = 0, preAcceptOwnValuesEnd = 0;
};

// The conflict indices of the n replicas, resident in HBM.
// epaxos/BallotHelpers.scala:11-21: (ordering, replicaIndex), compared lexicographically; nullBallot = (-1, -1)
struct Ballot {
  int32_t ordering = -1, replicaIndex = -1;
  static Ballot decode(int32_t e) { return e < 0 ? Ballot{} : Ballot{e >> 3, e & 7}; }
  bool operator==(const Ballot& o) const { return ordering == o.ordering && replicaIndex == o.replicaIndex; }
};
// Replica.cmdLog entry kinds (Replica.scala:303-330)
enum class EntryKind { None = 0, NoCommand = 1, PreAccepted = 2, Accepted = 3, Committed = 4 };
struct CmdLogEntry {
  EntryKind kind = EntryKind::None;
  Ballot ballot, voteBallot;
  int32_t tripleId = -1;               // the command, by the caller's id
  std::vector<int32_t> dependencies;   // the triple's dependency watermarks ({-1, ...}: known by tripleId only)
  int32_t ownValuesEnd = 0;
  Ballot largestBallot;                // of the replica that holds the entry (Replica.scala:458)
};
// a Prepare / Accept / PreAccept of one instance in one ballot, and the replicas it is delivered to
struct InstanceMessage {
  Instance instance;
  Ballot ballot;
  std::vector<int> recipients;
  int32_t tripleId = -1;                       // Accept, PreAccept
  Command command{-1, false};                  // Accept, PreAccept: the triple's command; key -1 = Noop
  std::vector<int32_t> dependencies;           // PreAccept: n watermarks (empty = none)
  int32_t ownValuesEnd = 0;
};
// what the recipients answered
struct InstanceReplies {
  std::vector<int> ok, resent, nacks, commits;           // replica indices
  Ballot nackBallot;                                     // the largest largestBallot a Nack carried
  bool committed = false;                                // Accept: f + 1 AcceptOks, the proposer's included
  std::vector<std::vector<int32_t>> replyDependencies;   // PreAccept: per replica, what its PreAcceptOk / Commit carried
  std::vector<int32_t> replyOwnValuesEnd, replyTripleId;
  std::vector<int32_t> replyStatus, replyVoteBallot;     // Prepare: per replica the PrepareOk's status (EntryKind, -1 = no
                                                         // PrepareOk) and voteBallot (encoded ordering * 8 + replicaIndex)
};
// Replica.handlePrepareOk's outcome for one instance (Replica.scala:1759-1884)
struct RecoveryDecision {
  enum Action { Wait = 0, AcceptPhase = 1, PreAcceptCommand = 2, PreAcceptNoop = 3 } action = Wait;
  int source = -1;          // the replica whose PrepareOk carries the triple / command (-1: none)
  int32_t tripleId = -1;
};

// The conflict indices (and, with numInstances > 0, the command logs) of the n replicas, resident in HBM.
class PreAcceptEngine {
 public:
  PreAcceptEngine(int f, int numKeys, int device = 0, int numInstances = 0) : n_(2 * f + 1) {
    if (f < 1) throw std::invalid_argument("f must be >= 1.");
    fpx_epx_config c{};
    c.num_replicas = n_;
    c.num_keys = numKeys;
    c.device = device;
    c.num_instances = numInstances;
    check(fpx_epx_create(&c, &epx_), "fpx_epx_create");
  }
  ~PreAcceptEngine() {
    if (epx_) fpx_epx_destroy(epx_);
  }
  PreAcceptEngine(const PreAcceptEngine&) = delete;
  PreAcceptEngine& operator=(const PreAcceptEngine&) = delete;
  int numReplicas() const { return n_; }

  // One tick.  deliveryOrder[r] = the order (indices into `proposals`) in which replica r processes the
  // tick's messages; empty => array order at every replica.
  std::vector<Decision> handleTick(const std::vector<Proposal>& proposals,
                                   const std::vector<std::vector<int>>& deliveryOrder = {}) {
    const int m = (int)proposals.size();
    std::vector<int32_t> leader(m), number(m), key(m), rank((size_t)n_ * m);
    std::vector<uint8_t> isSet(m), mask(m), seen(m);
    bool anyRecipients = false;
    for (int i = 0; i < m; ++i) {
      const Proposal& p = proposals[i];
      leader[i] = p.instance.replicaIndex, number[i] = p.instance.instanceNumber;
      key[i] = p.command.key, isSet[i] = p.command.isSet ? 1 : 0;
      unsigned b = 0;
      for (int r : p.fastQuorum) {
        if (r < 0 || r >= n_) throw std::invalid_argument("replica index out of range");
        b |= 1u << r;
      }
      mask[i] = (uint8_t)b;
      unsigned sb = p.recipients.empty() ? b : 0;
      for (int r : p.recipients) {
        if (r < 0 || r >= n_) throw std::invalid_argument("replica index out of range");
        sb |= 1u << r;
      }
      seen[i] = (uint8_t)sb;
      anyRecipients = anyRecipients || !p.recipients.empty();
    }
    if (!deliveryOrder.empty() && (int)deliveryOrder.size() != n_)
      throw std::invalid_argument("one delivery order per replica");
    for (int r = 0; r < n_; ++r) {
      if (deliveryOrder.empty()) {
        for (int i = 0; i < m; ++i) rank[(size_t)r * m + i] = i;
      } else {
        if ((int)deliveryOrder[r].size() != m) throw std::invalid_argument("a delivery order is a permutation");
        std::vector<char> seen(m, 0);
        for (int pos = 0; pos < m; ++pos) {
          const int i = deliveryOrder[r][pos];
          if (i < 0 || i >= m || seen[i]) throw std::invalid_argument("a delivery order is a permutation");
          seen[i] = 1;
          rank[(size_t)r * m + i] = pos;
        }
      }
    }
    std::vector<uint8_t> fast(m);
    std::vector<int32_t> deps((size_t)m * n_), ldeps((size_t)m * n_), own((size_t)m * 2);
    check(fpx_epx_preaccept(epx_, m, leader.data(), number.data(), key.data(), isSet.data(), mask.data(),
                            anyRecipients ? seen.data() : nullptr, rank.data(), nullptr, fast.data(), deps.data(), ldeps.data(),
                            own.data()),
          "Replica.handlePreAccept");
    std::vector<Decision> out(m);
    for (int i = 0; i < m; ++i) {
      out[i].fastPath = fast[i] != 0;
      out[i].dependencies.assign(deps.begin() + (size_t)i * n_, deps.begin() + (size_t)(i + 1) * n_);
      out[i].preAcceptDependencies.assign(ldeps.begin() + (size_t)i * n_, ldeps.begin() + (size_t)(i + 1) * n_);
      out[i].ownValuesEnd = own[(size_t)i * 2];
      out[i].preAcceptOwnValuesEnd = own[(size_t)i * 2 + 1];
    }
    return out;
  }

  // Replica.handlePrepare (Replica.scala:1632-1757), Replica.handlePreAccept in full (:1159-1289) and the Accept
  // phase (transitionToAcceptPhase :732-792, handleAccept :1421-1511, handleAcceptOk :1513-1565) on the command
  // log, one call per batch of pairwise distinct instances, messages delivered in vector order
  std::vector<InstanceReplies> handlePrepare(const std::vector<InstanceMessage>& msgs) { return run(0, msgs); }
  std::vector<InstanceReplies> acceptPhase(const std::vector<InstanceMessage>& msgs) { return run(1, msgs); }
  std::vector<InstanceReplies> handlePreAccept(const std::vector<InstanceMessage>& msgs) { return run(2, msgs); }
  // Replica.handlePrepareOk (Replica.scala:1759-1884): what the replica that sent prepares[i] does once it holds the
  // PrepareOks replies[i].ok (as handlePrepare returned them).  asIntended = false evaluates the reference's two tests
  // as written (:1810 compares a required enum with an Option, :1831 reads the Prepare's ballot: neither can hold)
  std::vector<RecoveryDecision> handlePrepareOks(const std::vector<InstanceMessage>& prepares,
                                                 const std::vector<InstanceReplies>& replies, bool asIntended = false) {
    const int m = (int)prepares.size();
    if (replies.size() != prepares.size()) throw std::invalid_argument("one set of replies per Prepare");
    std::vector<int32_t> leader(m), number(m), bo(m), br(m), rs((size_t)m * n_, -1), rv((size_t)m * n_, -1), rt((size_t)m * n_, -1);
    std::vector<uint8_t> mask(m);
    for (int i = 0; i < m; ++i) {
      leader[i] = prepares[i].instance.replicaIndex, number[i] = prepares[i].instance.instanceNumber;
      bo[i] = prepares[i].ballot.ordering, br[i] = prepares[i].ballot.replicaIndex;
      if ((int)replies[i].replyStatus.size() != n_) throw std::invalid_argument("replies of handlePrepare expected");
      unsigned bits = 0;
      for (int r : replies[i].ok) bits |= 1u << r;
      mask[i] = (uint8_t)bits;
      for (int r = 0; r < n_; ++r)
        rs[(size_t)i * n_ + r] = replies[i].replyStatus[r], rv[(size_t)i * n_ + r] = replies[i].replyVoteBallot[r],
                      rt[(size_t)i * n_ + r] = replies[i].replyTripleId[r];
    }
    std::vector<int32_t> act(m), src(m), tr(m);
    check(fpx_epx_handle_prepare_oks(epx_, m, leader.data(), number.data(), bo.data(), br.data(), mask.data(), rs.data(),
                                     rv.data(), rt.data(), asIntended ? 1 : 0, act.data(), src.data(), tr.data()),
          "Replica.handlePrepareOk");
    std::vector<RecoveryDecision> out(m);
    for (int i = 0; i < m; ++i) out[i].action = (RecoveryDecision::Action)act[i], out[i].source = src[i], out[i].tripleId = tr[i];
    return out;
  }

  CmdLogEntry cmdLog(int replica, Instance instance) {
    int32_t e[5];
    CmdLogEntry out;
    out.dependencies.assign(n_, 0);
    check(fpx_epx_read_cmdlog(epx_, replica, instance.replicaIndex, instance.instanceNumber, e), "cmdLog");
    check(fpx_epx_read_cmdlog_deps(epx_, replica, instance.replicaIndex, instance.instanceNumber, out.dependencies.data(),
                                   &out.ownValuesEnd), "cmdLog");
    out.kind = (EntryKind)e[0], out.ballot = Ballot::decode(e[1]), out.voteBallot = Ballot::decode(e[2]);
    out.tripleId = e[3], out.largestBallot = Ballot::decode(e[4]);
    return out;
  }

  // replica's conflict-index entry of a key: the TopOne watermarks of gets and of sets (util/TopOne.scala)
  std::pair<std::vector<int32_t>, std::vector<int32_t>> conflictIndex(int replica, int key) {
    std::vector<int32_t> g(n_), s2(n_);
    check(fpx_epx_read_index(epx_, replica, key, g.data(), s2.data()), "conflictIndex");
    return {g, s2};
  }

 private:
  static std::vector<int> replicasOf(unsigned bits) {
    std::vector<int> v;
    for (int r = 0; r < 8; ++r)
      if ((bits >> r) & 1u) v.push_back(r);
    return v;
  }
  // kind 0 Prepare, 1 Accept, 2 PreAccept
  std::vector<InstanceReplies> run(int kind, const std::vector<InstanceMessage>& msgs) {
    const int m = (int)msgs.size();
    std::vector<int32_t> leader(m), number(m), bo(m), br(m), tr(m), key(m), dend(m), din((size_t)m * n_, 0);
    std::vector<uint8_t> tgt(m), isSet(m);
    for (int i = 0; i < m; ++i) {
      const InstanceMessage& q = msgs[i];
      leader[i] = q.instance.replicaIndex, number[i] = q.instance.instanceNumber;
      bo[i] = q.ballot.ordering, br[i] = q.ballot.replicaIndex, tr[i] = q.tripleId;
      key[i] = q.command.key, isSet[i] = q.command.isSet ? 1 : 0, dend[i] = q.ownValuesEnd;
      unsigned bits = 0;
      for (int r : q.recipients) {
        if (r < 0 || r >= n_) throw std::invalid_argument("replica index out of range");
        bits |= 1u << r;
      }
      tgt[i] = (uint8_t)bits;
      if (!q.dependencies.empty()) {
        if ((int)q.dependencies.size() != n_) throw std::invalid_argument("one dependency watermark per replica");
        std::copy(q.dependencies.begin(), q.dependencies.end(), din.begin() + (size_t)i * n_);
      }
    }
    std::vector<uint8_t> ok(m), resend(m), nack(m), com(m), done(m);
    std::vector<int32_t> nb(m, -1), rd((size_t)m * n_ * n_), re((size_t)m * n_), rt((size_t)m * n_, -1);
    std::vector<int32_t> rs((size_t)m * n_, -1), rv((size_t)m * n_, -1);
    int32_t st;
    if (kind == 0)
      st = fpx_epx_prepare(epx_, m, leader.data(), number.data(), bo.data(), br.data(), tgt.data(), ok.data(), nack.data(),
                           com.data(), nb.data(), rs.data(), rv.data(), rt.data());
    else if (kind == 1)
      st = fpx_epx_accept(epx_, m, leader.data(), number.data(), bo.data(), br.data(), tr.data(), key.data(), isSet.data(),
                          tgt.data(), ok.data(), nack.data(), com.data(), nb.data(), done.data());
    else
      st = fpx_epx_handle_preaccept(epx_, m, leader.data(), number.data(), bo.data(), br.data(), key.data(), isSet.data(),
                                    tr.data(), din.data(), dend.data(), tgt.data(), ok.data(), resend.data(), nack.data(),
                                    com.data(), nb.data(), rd.data(), re.data(), rt.data());
    // a proposer that would have died in logger.fatal / logger.check (Replica.scala:740-757)
    if (st == FPX_EFATAL_PROTOCOL) throw std::logic_error("logger.fatal: the proposer's own command log refuses the Accept");
    check(st, kind == 0 ? "Replica.handlePrepare" : kind == 1 ? "Replica.handleAccept" : "Replica.handlePreAccept");
    std::vector<InstanceReplies> out(m);
    for (int i = 0; i < m; ++i) {
      out[i].ok = replicasOf(ok[i]), out[i].resent = replicasOf(resend[i]);
      out[i].nacks = replicasOf(nack[i]), out[i].commits = replicasOf(com[i]);
      out[i].nackBallot = Ballot::decode(nb[i]);
      out[i].committed = done[i] != 0;
      out[i].replyTripleId.assign(rt.begin() + (size_t)i * n_, rt.begin() + (size_t)(i + 1) * n_);
      if (kind == 0) {
        out[i].replyStatus.assign(rs.begin() + (size_t)i * n_, rs.begin() + (size_t)(i + 1) * n_);
        out[i].replyVoteBallot.assign(rv.begin() + (size_t)i * n_, rv.begin() + (size_t)(i + 1) * n_);
      }
      if (kind == 2) {
        out[i].replyOwnValuesEnd.assign(re.begin() + (size_t)i * n_, re.begin() + (size_t)(i + 1) * n_);
        for (int r = 0; r < n_; ++r)
          out[i].replyDependencies.emplace_back(rd.begin() + ((size_t)i * n_ + r) * n_,
                                                rd.begin() + ((size_t)i * n_ + r + 1) * n_);
      }
    }
    return out;
  }

 public:
  // Replica.handleCommit (Replica.scala:1567-1575 -> commit :815-830) at the replicas `at`: CommittedEntry(triple) whatever
  // their command logs held, the conflict index learns the command (key -1 = Noop).  dependencies: n watermarks (empty =
  // the triple is known by its id alone) + ownValuesEnd.
  void handleCommit(const Instance& instance, int tripleId, int key, bool isSet, const std::vector<int>& at,
                    const std::vector<int32_t>& dependencies = {}, int ownValuesEnd = 0) {
    if (!dependencies.empty() && (int)dependencies.size() != n_) throw std::invalid_argument("one watermark per replica");
    const int32_t L = instance.replicaIndex, x = instance.instanceNumber, tr = tripleId, k = key, end = ownValuesEnd;
    const uint8_t set = isSet ? 1 : 0;
    uint8_t mask = 0;
    for (int r : at) {
      if (r < 0 || r >= n_) throw std::invalid_argument("replica index out of range");
      mask |= (uint8_t)(1u << r);
    }
    check(fpx_epx_handle_commit(epx_, 1, &L, &x, &tr, &k, &set, dependencies.empty() ? nullptr : dependencies.data(),
                                dependencies.empty() ? nullptr : &end, &mask),
          "Replica.handleCommit");
  }

 private:
  int n_;
  fpx_epx* epx_ = nullptr;
};

// Replica.commit's tail + Replica.execute (epaxos/Replica.scala:859-917): every committed triple of a tick goes
// into the dependency graph (sequence number 0, :575-578), then appendExecute.  `committed[i]`: the instance of
// proposals[i] is committed -- a fast-path Decision, or a slow-path one whose Accept phase came back committed.
inline void commitToGraph(depgraph::DependencyGraph& graph, const std::vector<Proposal>& proposals,
                          const std::vector<Decision>& decisions, const std::vector<bool>& committed) {
  if (proposals.size() != decisions.size() || proposals.size() != committed.size())
    throw std::invalid_argument("one decision and one flag per proposal");
  const int n = graph.numLeaders();
  std::vector<int32_t> leader, id, deps, own;
  for (size_t i = 0; i < proposals.size(); ++i) {
    if (!committed[i]) continue;
    if ((int)decisions[i].dependencies.size() != n) throw std::invalid_argument("one watermark per replica");
    leader.push_back(proposals[i].instance.replicaIndex), id.push_back(proposals[i].instance.instanceNumber);
    deps.insert(deps.end(), decisions[i].dependencies.begin(), decisions[i].dependencies.end());
    own.push_back(decisions[i].ownValuesEnd);
  }
  graph.commitEpaxos(leader, id, deps, own);
}

}  // namespace epaxos
}  // namespace frankenpaxos
