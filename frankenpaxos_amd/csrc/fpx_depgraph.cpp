// fpx_depgraph.cpp -- dependency-graph execution (include/fpx_depgraph.h): host code of libfpx, no device code.
//
// The reference runs Tarjan's algorithm recursively over hash maps of boxed keys
// (depgraph/TarjanDependencyGraph.scala:358-462, depgraph/ZigzagTarjanDependencyGraph.scala:568-720).  Here: vertices in
// one flat pool addressed through a dense column per leader (id - base -> pool index), dependency watermarks in one
// int32 pool, the executed set as a watermark + bitmap per column, the metadata of one execute() stamped with an epoch
// instead of cleared, and strongConnect as an explicit frame stack -- a chain of a million dependent commands (one hot
// key) is a loop, not a million JVM stack frames.  Every step the reference takes is taken in the same order, so the
// executables, their order and the blockers are the same (the header says where the reference itself leaves the order
// open and what is taken there).
#include "../../include/fpx_depgraph.h"
#include "../../include/fpx.h"

#include <algorithm>
#include <cstring>
#include <new>
#include <set>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct Range {  // explicit ids [lo, hi) of column `leader` in a dependency set
  int32_t leader, lo, hi;
};

struct Vertex {
  int32_t leader, id, seq;
  // VertexMetadata (TarjanDependencyGraph.scala:189-195); valid iff epoch == Graph::epoch
  int32_t number, low_link, stack_index;
  uint32_t epoch;
  uint8_t eligible;
  uint8_t has_values;
  uint8_t live;
};

// IntPrefixSet (compact/IntPrefixSet.scala) of one column of the executed set: {0 .. wm-1} U bits
struct ExecSet {
  int32_t wm = 0;
  int64_t base_word = 0;  // bits[0] holds ids base_word*64 ..
  std::vector<uint64_t> bits;

  bool bit(int64_t x) const {
    int64_t w = (x >> 6) - base_word;
    return w >= 0 && w < (int64_t)bits.size() && ((bits[(size_t)w] >> (x & 63)) & 1u);
  }
  bool contains(int32_t x) const { return x < wm || bit(x); }
  // the first id in [from, to) that is not an explicit member (from >= wm), or `to`: a word at a time -- a column whose
  // watermark waits for one instance while thousands beyond it have executed is scanned by every dependent
  int32_t first_unset(int32_t from, int32_t to) const {
    int64_t x = from;
    while (x < to) {
      const int64_t w = (x >> 6) - base_word;
      if (w < 0 || w >= (int64_t)bits.size()) return (int32_t)x;
      const uint64_t zeros = ~bits[(size_t)w] >> (x & 63);  // bit k: id x + k is not a member
      if (zeros) {
        const int64_t y = x + __builtin_ctzll(zeros);
        return (int32_t)(y < to ? y : to);
      }
      x = (x | 63) + 1;
    }
    return to;
  }
  void compact() {  // :386-391
    for (;;) {
      const int64_t w = (wm >> 6) - base_word;
      if (w < 0 || w >= (int64_t)bits.size()) break;
      const uint64_t run = bits[(size_t)w] >> (wm & 63);  // members from wm on, within this word
      if (!(run & 1)) break;
      const int k = ~run ? __builtin_ctzll(~run) : 64 - (wm & 63);  // how many in a row
      const int take = k < 64 - (wm & 63) ? k : 64 - (wm & 63);
      const uint64_t mask = (take >= 64 ? ~0ull : ((1ull << take) - 1)) << (wm & 63);
      bits[(size_t)w] &= ~mask;
      wm += take;
    }
    // drop whole words below the watermark
    int64_t dead = (wm >> 6) - base_word;
    if (dead > 1024) {
      bits.erase(bits.begin(), bits.begin() + std::min<int64_t>(dead, (int64_t)bits.size()));
      base_word += dead;
    }
  }
  void add(int32_t x) {  // :185-196
    if (x < wm) return;
    int64_t w = (x >> 6) - base_word;
    if (w < 0) return;  // below the watermark by construction
    if (w >= (int64_t)bits.size()) bits.resize((size_t)w + 1 + (bits.size() >> 1), 0);
    bits[(size_t)w] |= 1ull << (x & 63);
    if (x == wm) compact();
  }
  void add_prefix(int32_t w) {  // addAll of a watermark-only set :264-301
    if (w <= wm) return;
    for (int32_t x = wm; x < w; x++) {  // forget explicit ids the new watermark covers
      int64_t k = (x >> 6) - base_word;
      if (k >= (int64_t)bits.size()) break;
      if (k >= 0) bits[(size_t)k] &= ~(1ull << (x & 63));
    }
    wm = w;
    compact();
  }
};

struct Column {  // util/BufferMap.scala: id -> vertex
  int32_t base = 0;
  std::vector<int32_t> slot;
  int32_t get(int32_t id) const {
    int64_t k = (int64_t)id - base;
    return (k < 0 || k >= (int64_t)slot.size()) ? -1 : slot[(size_t)k];
  }
};

struct Frame {
  int32_t v;
  // dependency iterator (InstancePrefixSet.diffIterator :118-126 over IntPrefixSet.DiffIterator :38-50)
  int32_t col, x, stage;
  uint32_t ri;
  // the dependency being looked at
  int32_t pl, pid;
  uint8_t has_pending;
  int32_t child;  // >= 0: a strongConnect(child) call has just returned
};

}  // namespace

struct fpx_depgraph {
  int32_t kind, L, gc_every;
  std::vector<Vertex> vtx;
  std::vector<int32_t> wm_pool;  // vtx index * L
  std::unordered_map<int32_t, std::vector<Range>> values;
  std::vector<int32_t> free_list;
  std::vector<Column> cols;
  std::vector<ExecSet> executed;
  std::vector<int32_t> executed_watermark;  // zigzag :296
  int64_t num_since_gc = 0, num_live = 0;
  // one execute()
  uint32_t epoch = 0;
  int32_t meta_count = 0;
  std::vector<int32_t> stack;
  std::vector<Frame> frames;
  std::vector<int32_t> out_leader, out_id, out_comp;
  std::set<std::pair<int32_t, int32_t>> blockers;
  std::vector<int32_t> scratch;

  int32_t alloc_vertex() {
    int32_t v;
    if (!free_list.empty()) {
      v = free_list.back();
      free_list.pop_back();
    } else {
      v = (int32_t)vtx.size();
      vtx.emplace_back();
      wm_pool.resize((size_t)(v + 1) * L);
    }
    return v;
  }
  void free_vertex(int32_t v) {
    if (vtx[v].has_values) values.erase(v);
    vtx[v].live = 0;
    vtx[v].has_values = 0;
    free_list.push_back(v);
    num_live--;
  }
  void put(int32_t leader, int32_t id, int32_t v) {  // BufferMap.put :37-51
    Column& c = cols[leader];
    int64_t k = (int64_t)id - c.base;
    if (k >= (int64_t)c.slot.size()) c.slot.resize((size_t)k + 1 + (c.slot.size() >> 1) + 16, -1);
    c.slot[(size_t)k] = v;
  }
  void emit(int32_t v) {
    out_leader.push_back(vtx[v].leader);
    out_id.push_back(vtx[v].id);
  }

  // the next dependency of f.v that is not in the executed set, looked for NOW
  bool next_dep(Frame& f, int32_t* l, int32_t* id) {
    const int32_t* wm = &wm_pool[(size_t)f.v * L];
    const std::vector<Range>* rs = nullptr;
    if (vtx[f.v].has_values) rs = &values[f.v];
    while (f.col < L) {
      const ExecSet& E = executed[f.col];
      if (f.stage == 0) {  // WatermarkIterator.getNext :124-145
        int32_t to = wm[f.col];
        if (f.x < to && to > E.wm) {
          int32_t start = E.first_unset(std::max(f.x, E.wm), to);
          if (start < to) {
            f.x = start + 1;
            *l = f.col;
            *id = start;
            return true;
          }
        }
        f.stage = 1;
        f.x = 0;
      }
      if (rs) {  // ValuesIterator.getNext :85-101, ascending
        while (f.ri < rs->size() && (*rs)[f.ri].leader < f.col) f.ri++;
        while (f.ri < rs->size() && (*rs)[f.ri].leader == f.col) {
          const Range& r = (*rs)[f.ri];
          int32_t x = std::max(f.x, r.lo);
          if (x < E.wm) x = E.wm;
          if (x < r.hi) x = E.first_unset(x, r.hi);
          if (x < r.hi) {
            f.x = x + 1;
            *l = f.col;
            *id = x;
            return true;
          }
          f.ri++;
          f.x = 0;
        }
      }
      f.col++;
      f.stage = 0;
      f.x = 0;
    }
    return false;
  }

  template <bool ZZ>
  void enter(int32_t v) {  // the head of strongConnect: Tarjan :366-374, zigzag :576-612
    Vertex& V = vtx[v];
    V.epoch = epoch;
    V.number = V.low_link = meta_count++;
    V.eligible = 1;
    Frame f{};
    f.v = v;
    f.child = -1;
    int32_t l = 0, id = 0;
    bool has = next_dep(f, &l, &id);
    if (ZZ && !has) {  // :583-599: nothing to wait for, executed on the spot
      V.stack_index = -1;
      emit(v);
      out_comp.push_back(1);
      executed[V.leader].add(V.id);
      return;
    }
    V.stack_index = (int32_t)stack.size();
    stack.push_back(v);
    f.has_pending = has;
    f.pl = l;
    f.pid = id;
    frames.push_back(f);
  }

  template <bool ZZ>
  void finish(int32_t v) {  // Tarjan :427-461, zigzag :686-719
    Vertex& V = vtx[v];
    if (V.low_link != V.number) return;
    if (V.stack_index == (int32_t)stack.size() - 1) {
      stack.pop_back();
      V.stack_index = -1;
      emit(v);
      out_comp.push_back(1);
      if (ZZ) executed[V.leader].add(V.id);
      return;
    }
    size_t from = (size_t)V.stack_index;
    scratch.assign(stack.begin() + from, stack.end());
    stack.resize(from);
    for (int32_t w : scratch) {
      vtx[w].stack_index = -1;
      if (ZZ) executed[vtx[w].leader].add(vtx[w].id);
    }
    std::sort(scratch.begin(), scratch.end(), [this](int32_t a, int32_t b) {
      const Vertex &A = vtx[a], &B = vtx[b];
      if (A.seq != B.seq) return A.seq < B.seq;
      if (A.leader != B.leader) return A.leader < B.leader;
      return A.id < B.id;
    });
    for (int32_t w : scratch) emit(w);
    out_comp.push_back((int32_t)scratch.size());
  }

  // strongConnect(root), iteratively; returns metadatas(root).eligible
  template <bool ZZ>
  bool strong_connect(int32_t root) {
    frames.clear();
    enter<ZZ>(root);
    while (!frames.empty()) {
      Frame& f = frames.back();
      int32_t v = f.v;
      if (f.child >= 0) {  // back from strongConnect(child): Tarjan :393-404, zigzag :637-650
        const Vertex& W = vtx[f.child];
        f.child = -1;
        if (!W.eligible) {
          vtx[v].eligible = 0;
          if (ZZ) vtx[v].stack_index = -1;
          frames.pop_back();
          continue;
        }
        vtx[v].low_link = std::min(vtx[v].low_link, W.low_link);
        f.has_pending = next_dep(f, &f.pl, &f.pid);
        continue;
      }
      if (!f.has_pending) {
        finish<ZZ>(v);
        frames.pop_back();
        continue;
      }
      int32_t w = cols[f.pl].get(f.pid);
      if (w < 0) {  // uncommitted child: Tarjan :380-389, zigzag :619-629
        vtx[v].eligible = 0;
        if (ZZ) vtx[v].stack_index = -1;
        blockers.insert({f.pl, f.pid});
        frames.pop_back();
        continue;
      }
      const Vertex& W = vtx[w];
      if (W.epoch != epoch) {  // unexplored child: recurse
        f.child = w;
        enter<ZZ>(w);  // may reallocate `frames`: f is dead from here
        continue;
      }
      if (!W.eligible) {  // Tarjan :406-413, zigzag :653-663
        vtx[v].eligible = 0;
        if (ZZ) vtx[v].stack_index = -1;
        frames.pop_back();
        continue;
      }
      if (W.stack_index != -1) vtx[v].low_link = std::min(vtx[v].low_link, W.number);  // on stack
      f.has_pending = next_dep(f, &f.pl, &f.pid);
    }
    return vtx[root].eligible != 0;
  }

  void begin_execute() {
    epoch++;
    if (epoch == 0) {  // wrapped: no stale stamp may look current
      for (Vertex& V : vtx) V.epoch = 0;
      epoch = 1;
    }
    meta_count = 0;
    stack.clear();
    out_leader.clear();
    out_id.clear();
    out_comp.clear();
    blockers.clear();
  }

  void execute_tarjan(int32_t num_blockers) {  // executeImpl :323-356, then the callers' bookkeeping :266-273
    bool stop = false;
    for (int32_t l = 0; l < L && !stop; l++) {
      Column& c = cols[l];
      for (size_t k = 0; k < c.slot.size(); k++) {
        int32_t v = c.slot[k];
        if (v < 0 || vtx[v].epoch == epoch) continue;
        if (!strong_connect<false>(v)) stack.clear();
        if (num_blockers >= 0 && (int64_t)blockers.size() >= num_blockers) {
          stop = true;
          break;
        }
      }
    }
    for (size_t i = 0; i < out_id.size(); i++) {
      int32_t l = out_leader[i], id = out_id[i];
      Column& c = cols[l];
      int32_t v = c.slot[(size_t)(id - c.base)];
      c.slot[(size_t)(id - c.base)] = -1;
      free_vertex(v);
      executed[l].add(id);
    }
  }

  bool execute_key_zigzag(int32_t l, int32_t id) {  // executeKeyImpl :502-566
    int32_t v = cols[l].get(id);
    if (v < 0) {
      blockers.insert({l, id});
      return false;
    }
    if (executed[l].contains(id)) return true;
    if (vtx[v].epoch != epoch) {
      if (!strong_connect<true>(v)) {
        for (int32_t u : stack) {
          vtx[u].eligible = 0;
          vtx[u].stack_index = -1;
        }
        stack.clear();
        return false;
      }
      return true;
    }
    return vtx[v].eligible != 0;
  }

  void execute_zigzag() {  // executeImpl :465-500, then :432-442
    std::vector<int32_t> eligible_columns((size_t)L);
    for (int32_t l = 0; l < L; l++) eligible_columns[(size_t)l] = l;
    size_t index = 0;
    while (!eligible_columns.empty()) {
      int32_t l = eligible_columns[index];
      if (execute_key_zigzag(l, executed_watermark[l])) {
        executed_watermark[l] = std::max(executed_watermark[l] + 1, executed[l].wm);
        index++;
        if (index >= eligible_columns.size()) index = 0;
      } else {
        eligible_columns.erase(eligible_columns.begin() + (long)index);
        if (index >= eligible_columns.size()) index = 0;
      }
    }
    num_since_gc += (int64_t)out_id.size();
    if (num_since_gc >= gc_every) {
      for (int32_t l = 0; l < L; l++) {  // BufferMap.garbageCollect :55-63
        Column& c = cols[l];
        int32_t w = executed_watermark[l];
        if (w <= c.base) continue;
        size_t drop = std::min((size_t)(w - c.base), c.slot.size());
        for (size_t k = 0; k < drop; k++)
          if (c.slot[k] >= 0) free_vertex(c.slot[k]);
        c.slot.erase(c.slot.begin(), c.slot.begin() + (long)drop);
        c.base = w;
      }
      num_since_gc = 0;
    }
  }

  // one vertex; vals = explicit ids as (leader, id) pairs, any order
  void commit_one(int32_t leader, int32_t id, int32_t seq, const int32_t* wm,
                  std::vector<std::pair<int32_t, int32_t>>& vals) {
    if (executed[leader].contains(id)) return;  // Tarjan :231, zigzag :350
    int32_t old = cols[leader].get(id);
    if (kind == FPX_DG_TARJAN) {
      if (old >= 0) return;  // "Ignore repeated commands" :231-236
    } else {
      if ((int64_t)id - cols[leader].base < 0) return;  // BufferMap.put below its watermark :40-42
      if (old >= 0) free_vertex(old);                  // replaced (header: quirk of :350)
    }
    int32_t v = alloc_vertex();
    Vertex& V = vtx[v];
    V.leader = leader;
    V.id = id;
    V.seq = seq;
    V.epoch = 0;
    V.live = 1;
    V.has_values = 0;
    int32_t* mywm = &wm_pool[(size_t)v * L];
    std::memcpy(mywm, wm, sizeof(int32_t) * (size_t)L);
    if (!vals.empty()) {  // IntPrefixSet's constructor compacts :161; ids under the watermark say nothing new
      std::sort(vals.begin(), vals.end());
      vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
      std::vector<Range> rs;
      for (auto& p : vals) {
        if (p.second < mywm[p.first]) continue;
        if (p.second == mywm[p.first]) {
          mywm[p.first]++;
          continue;
        }
        if (!rs.empty() && rs.back().leader == p.first && rs.back().hi == p.second)
          rs.back().hi++;
        else
          rs.push_back({p.first, p.second, p.second + 1});
      }
      if (!rs.empty()) {
        V.has_values = 1;
        values[v] = std::move(rs);
      }
    }
    put(leader, id, v);
    num_live++;
  }
};

extern "C" {

int32_t fpx_depgraph_create(const fpx_depgraph_config* cfg, fpx_depgraph** out) {
  if (!cfg || !out) return FPX_EINVAL;
  if ((cfg->kind != FPX_DG_TARJAN && cfg->kind != FPX_DG_ZIGZAG) || cfg->num_leaders < 1 || cfg->num_leaders > 4096)
    return FPX_EINVAL;
  fpx_depgraph* g = new (std::nothrow) fpx_depgraph();
  if (!g) return FPX_ENOMEM;
  g->kind = cfg->kind;
  g->L = cfg->num_leaders;
  g->gc_every = cfg->gc_every_n > 0 ? cfg->gc_every_n : 1000;
  g->cols.resize((size_t)g->L);
  g->executed.resize((size_t)g->L);
  g->executed_watermark.assign((size_t)g->L, 0);
  *out = g;
  return FPX_OK;
}

int32_t fpx_depgraph_destroy(fpx_depgraph* g) {
  delete g;
  return FPX_OK;
}

// The vertex columns and the executed sets are dense from their watermark on (one slot / one bit per id): an id far ahead
// of its column's executed watermark would buy hundreds of megabytes for ONE key (the reference's hash-based
// sets cost O(members)).  Such a key is refused -- FPX_ECAPACITY, nothing applied: a replica that far ahead of what it
// has executed has other problems (Replica.scala:859-917 commits what it is about to execute).
constexpr int64_t DG_MAX_AHEAD = (int64_t)1 << 26;
static int32_t check_keys(const fpx_depgraph* g, int32_t n, const int32_t* leader, const int32_t* id) {
  if (n < 0 || (n > 0 && (!leader || !id))) return FPX_EINVAL;
  for (int32_t i = 0; i < n; i++)
    if (leader[i] < 0 || leader[i] >= g->L || id[i] < 0) return FPX_EINVAL;
  for (int32_t i = 0; i < n; i++)
    if ((int64_t)id[i] - g->executed[(size_t)leader[i]].wm > DG_MAX_AHEAD) return FPX_ECAPACITY;
  return FPX_OK;
}

int32_t fpx_depgraph_commit(fpx_depgraph* g, int32_t n, const int32_t* leader, const int32_t* id, const int32_t* seq,
                            const int32_t* dep_watermark, const int64_t* dep_values_off,
                            const int32_t* dep_values_leader, const int32_t* dep_values_id) {
  if (!g) return FPX_EINVAL;
  int32_t st = check_keys(g, n, leader, id);
  if (st) return st;
  if (n > 0 && !dep_watermark) return FPX_EINVAL;
  for (int64_t k = 0; k < (int64_t)n * g->L; k++)
    if (dep_watermark[k] < 0) return FPX_EINVAL;
  if (dep_values_off) {
    if (dep_values_off[0] < 0) return FPX_EINVAL;
    for (int32_t i = 0; i < n; i++)
      if (dep_values_off[i + 1] < dep_values_off[i]) return FPX_EINVAL;
    if (dep_values_off[n] > dep_values_off[0] && (!dep_values_leader || !dep_values_id)) return FPX_EINVAL;
    for (int64_t j = dep_values_off[0]; j < dep_values_off[n]; j++)
      if (dep_values_leader[j] < 0 || dep_values_leader[j] >= g->L || dep_values_id[j] < 0) return FPX_EINVAL;
  }
  try {
    std::vector<std::pair<int32_t, int32_t>> vals;
    for (int32_t i = 0; i < n; i++) {
      vals.clear();
      if (dep_values_off)
        for (int64_t j = dep_values_off[i]; j < dep_values_off[i + 1]; j++)
          vals.push_back({dep_values_leader[j], dep_values_id[j]});
      g->commit_one(leader[i], id[i], seq ? seq[i] : 0, dep_watermark + (size_t)i * g->L, vals);
    }
  } catch (const std::bad_alloc&) {
    return FPX_ENOMEM;
  }
  return FPX_OK;
}

int32_t fpx_depgraph_commit_epx(fpx_depgraph* g, int32_t n, const int32_t* leader, const int32_t* id,
                                const int32_t* seq, const int32_t* deps, const int32_t* own_values_end,
                                int32_t own_stride, const uint8_t* mask) {
  if (!g) return FPX_EINVAL;
  int32_t st = check_keys(g, n, leader, id);
  if (st) return st;
  if (n > 0 && !deps) return FPX_EINVAL;
  if (own_values_end && own_stride < 1) return FPX_EINVAL;
  for (int32_t i = 0; i < n; i++) {
    if (mask && !mask[i]) continue;
    for (int32_t l = 0; l < g->L; l++)
      if (deps[(size_t)i * g->L + l] < 0) return FPX_EINVAL;
    if (own_values_end && own_values_end[(size_t)i * own_stride] < 0) return FPX_EINVAL;
  }
  try {
    std::vector<std::pair<int32_t, int32_t>> vals;
    for (int32_t i = 0; i < n; i++) {
      if (mask && !mask[i]) continue;
      vals.clear();
      int32_t end = own_values_end ? own_values_end[(size_t)i * own_stride] : 0;
      for (int32_t x = id[i] + 1; x < end; x++) vals.push_back({leader[i], x});
      g->commit_one(leader[i], id[i], seq ? seq[i] : 0, deps + (size_t)i * g->L, vals);
    }
  } catch (const std::bad_alloc&) {
    return FPX_ENOMEM;
  }
  return FPX_OK;
}

int32_t fpx_depgraph_update_executed(fpx_depgraph* g, const int32_t* watermark, int32_t n, const int32_t* leader,
                                     const int32_t* id) {
  if (!g) return FPX_EINVAL;
  int32_t st = check_keys(g, n, leader, id);
  if (st) return st;
  if (watermark)
    for (int32_t l = 0; l < g->L; l++)
      if (watermark[l] < 0) return FPX_EINVAL;
  try {
    if (watermark)
      for (int32_t l = 0; l < g->L; l++) g->executed[(size_t)l].add_prefix(watermark[l]);
    for (int32_t i = 0; i < n; i++) g->executed[(size_t)leader[i]].add(id[i]);
    if (g->kind == FPX_DG_TARJAN) {  // vertices.retain(!executed.contains) :243
      for (int32_t l = 0; l < g->L; l++) {
        Column& c = g->cols[(size_t)l];
        for (size_t k = 0; k < c.slot.size(); k++)
          if (c.slot[k] >= 0 && g->executed[(size_t)l].contains(c.base + (int32_t)k)) {
            g->free_vertex(c.slot[k]);
            c.slot[k] = -1;
          }
      }
    }
  } catch (const std::bad_alloc&) {
    return FPX_ENOMEM;
  }
  return FPX_OK;
}

int32_t fpx_depgraph_execute(fpx_depgraph* g, int32_t num_blockers, int64_t* num_executables, int64_t* num_components,
                             int64_t* num_blockers_found) {
  if (!g) return FPX_EINVAL;
  try {
    g->begin_execute();
    if (g->kind == FPX_DG_TARJAN)
      g->execute_tarjan(num_blockers);
    else
      g->execute_zigzag();
  } catch (const std::bad_alloc&) {
    return FPX_ENOMEM;
  }
  if (num_executables) *num_executables = (int64_t)g->out_id.size();
  if (num_components) *num_components = (int64_t)g->out_comp.size();
  if (num_blockers_found) *num_blockers_found = (int64_t)g->blockers.size();
  return FPX_OK;
}

int32_t fpx_depgraph_read_result(fpx_depgraph* g, int32_t* exec_leader, int32_t* exec_id, int32_t* component_size,
                                 int32_t* blocker_leader, int32_t* blocker_id) {
  if (!g) return FPX_EINVAL;
  size_t n = g->out_id.size();
  if (exec_leader && n) std::memcpy(exec_leader, g->out_leader.data(), n * sizeof(int32_t));
  if (exec_id && n) std::memcpy(exec_id, g->out_id.data(), n * sizeof(int32_t));
  if (component_size && !g->out_comp.empty())
    std::memcpy(component_size, g->out_comp.data(), g->out_comp.size() * sizeof(int32_t));
  size_t k = 0;
  for (const auto& b : g->blockers) {
    if (blocker_leader) blocker_leader[k] = b.first;
    if (blocker_id) blocker_id[k] = b.second;
    k++;
  }
  return FPX_OK;
}

int64_t fpx_depgraph_num_vertices(fpx_depgraph* g) { return g ? g->num_live : -1; }

int32_t fpx_depgraph_executed_watermark(fpx_depgraph* g, int32_t* watermark) {
  if (!g || !watermark) return FPX_EINVAL;
  for (int32_t l = 0; l < g->L; l++) watermark[l] = g->executed[(size_t)l].wm;
  return FPX_OK;
}

}  // extern "C"
