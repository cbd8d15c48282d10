// fpx.hpp -- C++ host-side mirror of the reference's interface for the Phase-2 path, on top of the
// C ABI of include/fpx.h.  The reference is Scala (no JDK in this image), so this is the host
// language layer a user of the reference would program against: same names, argument meaning and
// error behaviour as
//
//   frankenpaxos.roundsystem.RoundSystem.ClassicRoundRobin   roundsystem/RoundSystem.scala:60-87
//   frankenpaxos.quorums.{QuorumSystem,SimpleMajority,Grid,UnanimousWrites}  quorums/*.scala
//   frankenpaxos.multipaxos.Acceptor.handlePhase2a           multipaxos/Acceptor.scala:184-220
//   frankenpaxos.multipaxos.ProxyLeader.handlePhase2a/2b     multipaxos/ProxyLeader.scala:175-258
//
// but batched: a handler takes the messages one event-loop tick delivered and returns the messages
// the reference handlers would have sent.  require(...) failures throw std::invalid_argument
// (IllegalArgumentException); logger.fatal throws std::logic_error.  Everything computes on the GPU
// through libfpx; there is no host implementation of the predicates or handlers in this file.
#pragma once

#include <cstdint>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fpx.h"

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
};

}  // namespace multipaxos
}  // namespace frankenpaxos
