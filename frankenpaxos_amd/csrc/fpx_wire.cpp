// fpx_wire.cpp -- the wire adapter of include/fpx_wire.h: protobuf (proto2, ScalaPB-compatible canonical
// encoding) <-> SoA batches for the messages of the Phase-2 path.  Host code, no dependencies: a varint reader /
// writer and one small parser per message of shared/src/main/scala/frankenpaxos/multipaxos/MultiPaxos.proto,
// mencius/Mencius.proto and epaxos/EPaxos.proto (+ compact/IntPrefixSet.proto).
#include "../../include/fpx_wire.h"

#include <cstring>
#include <vector>

#include "../../include/fpx.h"
#include "fpx_wire_parse.hpp"

namespace {

using namespace fpxw;

// ---- writing ---------------------------------------------------------------------------------------------
struct Writer {
  uint8_t* p;        // null: only count
  int64_t n = 0;
  void byte(uint8_t b) {
    if (p) p[n] = b;
    ++n;
  }
  void varint(uint64_t v) {
    while (v >= 0x80) {
      byte((uint8_t)(v | 0x80));
      v >>= 7;
    }
    byte((uint8_t)v);
  }
  void tag(uint32_t field, uint32_t wt) { varint(((uint64_t)field << 3) | wt); }
  void i32(uint32_t field, int32_t v) {
    tag(field, 0);
    varint((uint64_t)(int64_t)v);  // negative: sign-extended, 10 bytes (protobuf int32)
  }
  void bytes(const uint8_t* src, int64_t len) {
    if (p && len) memcpy(p + n, src, (size_t)len);
    n += len;
  }
};

int64_t varint_len(uint64_t v) {
  int64_t k = 1;
  while (v >= 0x80) v >>= 7, ++k;
  return k;
}
int64_t i32_len(int32_t v) { return 1 + varint_len((uint64_t)(int64_t)v); }  // fields 1..15: one tag byte

// the CommandBatchOrNoop body to embed: the caller's bytes, or {noop = 2: empty Noop} = 12 00
static const uint8_t NOOP_VALUE[2] = {0x12, 0x00};
inline void pick_value(const uint8_t*& value, int32_t& len, int32_t is_noop) {
  if (is_noop) value = NOOP_VALUE, len = 2;
  if (len < 0) len = 0;
}

// wraps `inner_len` bytes produced by `emit` as field `wrapper_field` (length-delimited) of an ...Inbound message
template <typename F>
int64_t wrapped(uint8_t* out, int64_t cap, uint32_t wrapper_field, int64_t inner_len, F emit) {
  const int64_t total = 1 + varint_len((uint64_t)inner_len) + inner_len;
  if (total > cap || !out) return -total;
  Writer w{out};
  w.tag(wrapper_field, 2);
  w.varint((uint64_t)inner_len);
  emit(w);
  return w.n;
}

int64_t encode_phase2a(uint8_t* out, int64_t cap, uint32_t wrapper_field, int32_t slot, int32_t round,
                       const uint8_t* value, int32_t value_len, int32_t is_noop) {
  pick_value(value, value_len, is_noop);
  const int64_t inner = i32_len(slot) + i32_len(round) + 1 + varint_len((uint64_t)value_len) + value_len;
  return wrapped(out, cap, wrapper_field, inner, [&](Writer& w) {
    w.i32(1, slot);
    w.i32(2, round);
    w.tag(3, 2);
    w.varint((uint64_t)value_len);
    w.bytes(value, value_len);
  });
}

int64_t phase2b_len(int32_t g, int32_t a, int32_t slot, int32_t round) {
  return i32_len(g) + i32_len(a) + i32_len(slot) + i32_len(round);
}

template <typename Emit>
int32_t decode_loop(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n, int32_t* bad_index,
                    Emit emit) {
  if (n < 0 || buf_len < 0 || (n > 0 && (!buf || !offsets))) return FPX_EINVAL;
  // every offset is checked against the buffer BEFORE a single byte is parsed: offsets like [0, 10^9, 5] must not
  // send message 0's parser a gigabyte past the end before the non-monotone pair is noticed
  for (int32_t i = 0; i <= n && n > 0; ++i) {
    if (!offset_ok(offsets, i, buf_len)) {
      if (bad_index) *bad_index = i < n ? i : n - 1;
      return FPX_EINVAL;
    }
  }
  for (int32_t i = 0; i < n; ++i) {
    bool ok = true;
    if (ok) {
      Reader r{buf + offsets[i], buf + offsets[i + 1]};
      ok = emit(i, r);
    }
    if (!ok) {
      if (bad_index) *bad_index = i;
      return FPX_EINVAL;
    }
  }
  return FPX_OK;
}

}  // namespace

extern "C" {

int32_t fpx_wire_decode_proxy_leader_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n, int32_t* kind,
                                             int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                             int32_t* value_len, int32_t* group_index, int32_t* acceptor_index,
                                             int32_t* bad_index) {
  if (n > 0 && (!kind || !slot || !round)) return FPX_EINVAL;
  return decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    Msg o;  // the parser itself is fpx_wire_parse.hpp's, shared with the device decoder
    const bool ok = parse_proxy_leader_inbound(buf, r, &o);
    kind[i] = o.kind, slot[i] = o.slot, round[i] = o.round;
    if (is_noop) is_noop[i] = o.is_noop;
    if (value_off) value_off[i] = o.value_off;
    if (value_len) value_len[i] = o.value_len;
    if (group_index) group_index[i] = o.a;
    if (acceptor_index) acceptor_index[i] = o.b;
    return ok;
  });
}

int32_t fpx_wire_decode_acceptor_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n, int32_t* kind,
                                         int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                         int32_t* value_len, int32_t* chosen_watermark, int32_t* bad_index) {
  if (n > 0 && (!kind || !slot || !round)) return FPX_EINVAL;
  return decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    Msg o;
    const bool ok = parse_acceptor_inbound(buf, r, &o);
    kind[i] = o.kind, slot[i] = o.slot, round[i] = o.round;
    if (is_noop) is_noop[i] = o.is_noop;
    if (value_off) value_off[i] = o.value_off;
    if (value_len) value_len[i] = o.value_len;
    if (chosen_watermark) chosen_watermark[i] = o.a;
    return ok;
  });
}

int32_t fpx_wire_decode_replica_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n, int32_t* kind,
                                        int32_t* slot, int32_t* is_noop, int64_t* value_off, int32_t* value_len,
                                        int32_t* bad_index) {
  if (n > 0 && (!kind || !slot)) return FPX_EINVAL;
  return decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    kind[i] = FPX_WIRE_OTHER, slot[i] = -1;
    if (is_noop) is_noop[i] = -1;
    if (value_off) value_off[i] = -1;
    if (value_len) value_len[i] = -1;
    while (r.more()) {
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field == 1 && wt == 2) {  // Chosen
        Fields f;
        if (!parse_flat(r.sub(), 2, &f) || !r.ok || (f.seen & 0x2) != 0x2 || !f.has_value) return false;
        kind[i] = FPX_WIRE_CHOSEN, slot[i] = f.i[1];
        if (is_noop) is_noop[i] = f.value.is_noop;
        if (value_off) value_off[i] = f.value.at - buf;
        if (value_len) value_len[i] = f.value.len;
      } else {
        r.skip(wt);
      }
    }
    return r.ok;
  });
}

int32_t fpx_wire_phase2b_rows(int32_t n, const int32_t* kind, const int32_t* group_index,
                              const int32_t* acceptor_index, const int32_t* slot, const int32_t* round,
                              int32_t grid_cols, int32_t* num_rows, int32_t* row_slot, int32_t* row_round,
                              uint64_t* row_bits) {
  if (n < 0 || !num_rows || (n > 0 && (!kind || !group_index || !acceptor_index || !slot || !round || !row_slot ||
                                       !row_round || !row_bits)))
    return FPX_EINVAL;
  // (slot, round) -> row: open addressing over a power-of-two table at most half full, linear probing; the votes of
  // one slot usually arrive next to each other, so the previous message's row is tried first
  size_t cap = 16;
  while (cap < (size_t)n * 2) cap <<= 1;
  // row index per table entry (the key is read back from row_slot / row_round).  The table lives across calls (one per
  // calling thread) and only the entries a call used are reset: a tick's worth of fresh pages per call would cost more
  // than the fold itself
  static thread_local std::vector<int32_t> table;
  static thread_local std::vector<uint32_t> used;
  if (table.size() < cap) table.assign(cap, -1);
  used.clear();
  const size_t mask = cap - 1;
  int32_t rows = 0, prev = -1;
  for (int32_t i = 0; i < n; ++i) {
    if (kind[i] != FPX_WIRE_PHASE2B) continue;
    const int64_t bit = grid_cols > 0 ? (int64_t)group_index[i] * grid_cols + acceptor_index[i] : acceptor_index[i];
    if (bit < 0 || bit >= FPX_MAX_REPLICAS || acceptor_index[i] < 0 || (grid_cols > 0 && acceptor_index[i] >= grid_cols)) {
      for (uint32_t at : used) table[at] = -1;
      return FPX_EINVAL;
    }
    int32_t row = -1;
    if (prev >= 0 && row_slot[prev] == slot[i] && row_round[prev] == round[i]) {
      row = prev;
    } else {
      uint64_t h = ((uint64_t)(uint32_t)slot[i] << 32) | (uint32_t)round[i];
      h ^= h >> 33, h *= 0xff51afd7ed558ccdull, h ^= h >> 33;  // the slot sits in the high half: fold it down first
      size_t at = (size_t)h & mask;
      for (;; at = (at + 1) & mask) {
        const int32_t r = table[at];
        if (r < 0) break;
        if (row_slot[r] == slot[i] && row_round[r] == round[i]) {
          row = r;
          break;
        }
      }
      if (row < 0) {
        row = rows++;
        table[at] = row;
        used.push_back((uint32_t)at);
        row_slot[row] = slot[i], row_round[row] = round[i];
        memset(row_bits + (size_t)row * 4, 0, 32);
      }
    }
    prev = row;
    row_bits[(size_t)row * 4 + (bit >> 6)] |= 1ull << (bit & 63);
  }
  for (uint32_t at : used) table[at] = -1;
  *num_rows = rows;
  return FPX_OK;
}

int64_t fpx_wire_encode_proxy_leader_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round,
                                             const uint8_t* value, int32_t value_len, int32_t is_noop) {
  return encode_phase2a(out, cap, 1, slot, round, value, value_len, is_noop);
}

int64_t fpx_wire_encode_acceptor_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round, const uint8_t* value,
                                         int32_t value_len, int32_t is_noop) {
  return encode_phase2a(out, cap, 2, slot, round, value, value_len, is_noop);
}

int64_t fpx_wire_encode_acceptor_phase1a(uint8_t* out, int64_t cap, int32_t round, int32_t chosen_watermark) {
  return wrapped(out, cap, 1, i32_len(round) + i32_len(chosen_watermark), [&](Writer& w) {
    w.i32(1, round);
    w.i32(2, chosen_watermark);
  });
}

int64_t fpx_wire_encode_proxy_leader_phase2b(uint8_t* out, int64_t cap, int32_t group_index, int32_t acceptor_index,
                                             int32_t slot, int32_t round) {
  return wrapped(out, cap, 2, phase2b_len(group_index, acceptor_index, slot, round), [&](Writer& w) {
    w.i32(1, group_index);
    w.i32(2, acceptor_index);
    w.i32(3, slot);
    w.i32(4, round);
  });
}

int64_t fpx_wire_encode_replica_chosen(uint8_t* out, int64_t cap, int32_t slot, const uint8_t* value,
                                       int32_t value_len, int32_t is_noop) {
  pick_value(value, value_len, is_noop);
  const int64_t inner = i32_len(slot) + 1 + varint_len((uint64_t)value_len) + value_len;
  return wrapped(out, cap, 1, inner, [&](Writer& w) {
    w.i32(1, slot);
    w.tag(2, 2);
    w.varint((uint64_t)value_len);
    w.bytes(value, value_len);
  });
}

int64_t fpx_wire_encode_leader_nack(uint8_t* out, int64_t cap, int32_t round) {
  return wrapped(out, cap, 6, i32_len(round), [&](Writer& w) { w.i32(1, round); });
}

int64_t fpx_wire_encode_client_max_slot_reply(uint8_t* out, int64_t cap, const uint8_t* command_id, int32_t command_id_len,
                                              int32_t group_index, int32_t acceptor_index, int32_t slot) {
  if (command_id_len < 0 || (command_id_len > 0 && !command_id)) return 0;
  const int64_t inner = 1 + varint_len((uint64_t)command_id_len) + command_id_len + i32_len(group_index) + i32_len(acceptor_index) + i32_len(slot);
  return wrapped(out, cap, 4, inner, [&](Writer& w) {
    w.tag(1, 2);
    w.varint((uint64_t)command_id_len);
    w.bytes(command_id, command_id_len);
    w.i32(2, group_index);
    w.i32(3, acceptor_index);
    w.i32(4, slot);
  });
}

int64_t fpx_wire_encode_read_batcher_batch_max_slot_reply(uint8_t* out, int64_t cap, int32_t read_batcher_index,
                                                          int32_t read_batcher_id, int32_t acceptor_index, int32_t slot) {
  const int64_t inner = i32_len(read_batcher_index) + i32_len(read_batcher_id) + i32_len(acceptor_index) + i32_len(slot);
  return wrapped(out, cap, 4, inner, [&](Writer& w) {
    w.i32(1, read_batcher_index);
    w.i32(2, read_batcher_id);
    w.i32(3, acceptor_index);
    w.i32(4, slot);
  });
}

int64_t fpx_wire_encode_leader_phase1b(uint8_t* out, int64_t cap, int32_t group_index, int32_t acceptor_index,
                                       int32_t round, int32_t n_info, const int32_t* slot, const int32_t* vote_round,
                                       const uint8_t* values, const int64_t* value_off, const int32_t* value_len,
                                       const uint8_t* is_noop) {
  if (n_info < 0 || (n_info > 0 && (!slot || !vote_round))) return 0;
  auto value_of = [&](int32_t j, const uint8_t*& v, int32_t& len) {
    const bool noop = is_noop && is_noop[j];
    v = (!noop && values && value_off) ? values + value_off[j] : nullptr;
    len = (!noop && value_len) ? value_len[j] : 0;
    pick_value(v, len, noop ? 1 : 0);
  };
  auto info_len = [&](int32_t j) {
    const uint8_t* v;
    int32_t len;
    value_of(j, v, len);
    return i32_len(slot[j]) + i32_len(vote_round[j]) + 1 + varint_len((uint64_t)len) + len;
  };
  int64_t inner = i32_len(group_index) + i32_len(acceptor_index) + i32_len(round);
  for (int32_t j = 0; j < n_info; ++j) {
    const int64_t il = info_len(j);
    inner += 1 + varint_len((uint64_t)il) + il;
  }
  return wrapped(out, cap, 1, inner, [&](Writer& w) {
    w.i32(1, group_index);
    w.i32(2, acceptor_index);
    w.i32(3, round);
    for (int32_t j = 0; j < n_info; ++j) {
      const uint8_t* v;
      int32_t len;
      value_of(j, v, len);
      w.tag(4, 2);
      w.varint((uint64_t)info_len(j));
      w.i32(1, slot[j]);
      w.i32(2, vote_round[j]);
      w.tag(3, 2);
      w.varint((uint64_t)len);
      w.bytes(v, len);
    }
  });
}

int32_t fpx_wire_decode_leader_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                       int32_t* kind, int32_t* round, int32_t* group_index, int32_t* acceptor_index,
                                       int32_t* info_first, int32_t* info_count, int32_t info_cap, int32_t* info_total,
                                       int32_t* info_slot, int32_t* info_vote_round, int32_t* info_is_noop,
                                       int64_t* info_value_off, int32_t* info_value_len, int32_t* bad_index) {
  if ((n > 0 && (!kind || !round)) || info_cap < 0 || (info_cap > 0 && (!info_slot || !info_vote_round))) return FPX_EINVAL;
  int32_t total = 0;
  if (info_total) *info_total = 0;
  const int32_t st = decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    kind[i] = FPX_WIRE_OTHER, round[i] = -1;
    if (group_index) group_index[i] = -1;
    if (acceptor_index) acceptor_index[i] = -1;
    if (info_first) info_first[i] = total;
    if (info_count) info_count[i] = 0;
    while (r.more()) {  // LeaderInbound: the last member of the oneof that is present wins
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field == 1 && wt == 2) {  // Phase1b
        Reader p = r.sub();
        if (!r.ok) return false;
        int32_t hdr[4] = {0, 0, 0, 0};
        unsigned seen = 0;
        const int32_t first = total;  // a second phase1b member replaces the first one's entries
        while (p.more()) {
          const uint64_t t2 = p.varint();
          const uint32_t f2 = (uint32_t)(t2 >> 3), w2 = (uint32_t)(t2 & 7);
          if (f2 >= 1 && f2 <= 3 && w2 == 0) {
            hdr[f2] = as_i32(p.varint()), seen |= 1u << f2;
          } else if (f2 == 4 && w2 == 2) {
            Fields f;
            if (!parse_flat(p.sub(), 3, &f) || !p.ok || (f.seen & 0x6) != 0x6 || !f.has_value) return false;
            if (total >= info_cap) return false;
            info_slot[total] = f.i[1], info_vote_round[total] = f.i[2];
            if (info_is_noop) info_is_noop[total] = f.value.is_noop;
            if (info_value_off) info_value_off[total] = f.value.at - buf;
            if (info_value_len) info_value_len[total] = f.value.len;
            ++total;
          } else {
            p.skip(w2);
          }
        }
        if (!p.ok || (seen & 0xe) != 0xe) return false;
        kind[i] = FPX_WIRE_PHASE1B, round[i] = hdr[3];
        if (group_index) group_index[i] = hdr[1];
        if (acceptor_index) acceptor_index[i] = hdr[2];
        if (info_first) info_first[i] = first;
        if (info_count) info_count[i] = total - first;
      } else if (field == 6 && wt == 2) {  // Nack
        Fields f;
        if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x2) != 0x2) return false;
        kind[i] = FPX_WIRE_NACK, round[i] = f.i[1];
      } else {
        r.skip(wt);  // ClientRequest, ChosenWatermark, Recover, ...: the JVM Leader's own
      }
    }
    return r.ok;
  });
  if (info_total) *info_total = total;
  return st;
}

int64_t fpx_wire_encode_phase2b_batch(int32_t n, const int32_t* slot, const int32_t* round, const uint64_t* vote_bits,
                                      const int32_t* group_of_slot, int32_t grid_cols, uint8_t* out, int64_t cap,
                                      int64_t* out_offsets, int64_t max_msgs) {
  if (n < 0 || (n > 0 && (!slot || !round || !vote_bits)) || !out || !out_offsets) return -1;
  int64_t count = 0, at = 0;
  out_offsets[0] = 0;
  for (int32_t i = 0; i < n; ++i) {
    for (int w = 0; w < 4; ++w) {
      uint64_t x = vote_bits[(size_t)i * 4 + w];
      while (x) {
        const int bit = w * 64 + __builtin_ctzll(x);
        x &= x - 1;
        const int32_t g = grid_cols > 0 ? bit / grid_cols : (group_of_slot ? group_of_slot[i] : 0);
        const int32_t a = grid_cols > 0 ? bit % grid_cols : bit;
        if (count >= max_msgs) return -1;
        const int64_t len = fpx_wire_encode_proxy_leader_phase2b(out + at, cap - at, g, a, slot[i], round[i]);
        if (len < 0) return -1;
        at += len;
        out_offsets[++count] = at;
      }
    }
  }
  return count;
}


// ---- Mencius (mencius/Mencius.proto) ---------------------------------------------------------------------------
namespace {

struct MenciusOut {
  int32_t *kind, *slot, *slot_end, *round, *is_noop;
  int64_t* value_off;
  int32_t *value_len, *group_index, *acceptor_index, *chosen_watermark;
  const uint8_t* buf;
  void clear(int32_t i) const {
    kind[i] = FPX_WIRE_OTHER, slot[i] = -1;
    if (slot_end) slot_end[i] = -1;
    if (round) round[i] = -1;
    if (is_noop) is_noop[i] = -1;
    if (value_off) value_off[i] = -1;
    if (value_len) value_len[i] = -1;
    if (group_index) group_index[i] = -1;
    if (acceptor_index) acceptor_index[i] = -1;
    if (chosen_watermark) chosen_watermark[i] = -1;
  }
  // Phase2a { slot = 1; round = 2; command_batch_or_noop = 3 }  /  Chosen { slot = 1; command_batch_or_noop = 2 }
  bool with_value(int32_t i, Reader& r, int value_field, int32_t k) const {
    Fields f;
    const unsigned need = value_field == 3 ? 0x6u : 0x2u;
    if (!parse_flat(r.sub(), value_field, &f) || !r.ok || (f.seen & need) != need || !f.has_value) return false;
    clear(i);  // a oneof: the last member wins, and nothing of an earlier member stays behind
    kind[i] = k, slot[i] = f.i[1];
    if (round && value_field == 3) round[i] = f.i[2];
    if (is_noop) is_noop[i] = f.value.is_noop;
    if (value_off) value_off[i] = f.value.at - buf;
    if (value_len) value_len[i] = f.value.len;
    return true;
  }
  // Phase2aNoopRange { start = 1; end = 2; round = 3 }  /  ChosenNoopRange { start = 1; end = 2 }
  bool range(int32_t i, Reader& r, bool has_round, int32_t k) const {
    Fields f;
    const unsigned need = has_round ? 0xeu : 0x6u;
    if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & need) != need) return false;
    clear(i);
    kind[i] = k, slot[i] = f.i[1];
    if (slot_end) slot_end[i] = f.i[2];
    if (round && has_round) round[i] = f.i[3];
    return true;
  }
};

}  // namespace

int32_t fpx_wire_mencius_decode_proxy_leader_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets,
                                                     int32_t n, int32_t* kind, int32_t* slot, int32_t* slot_end,
                                                     int32_t* round, int32_t* is_noop, int64_t* value_off,
                                                     int32_t* value_len, int32_t* group_index, int32_t* acceptor_index,
                                                     int32_t* bad_index) {
  if (n > 0 && (!kind || !slot || !round)) return FPX_EINVAL;
  const MenciusOut o{kind, slot, slot_end, round, is_noop, value_off, value_len, group_index, acceptor_index, nullptr, buf};
  return decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    o.clear(i);
    while (r.more()) {
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field == 2 && wt == 2) {
        if (!o.with_value(i, r, 3, FPX_WIRE_PHASE2A)) return false;
      } else if (field == 3 && wt == 2) {
        if (!o.range(i, r, true, FPX_WIRE_PHASE2A_NOOP_RANGE)) return false;
      } else if (field == 4 && wt == 2) {  // Phase2b { acceptor_index = 1; slot = 2; round = 3 }
        Fields f;
        if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0xe) != 0xe) return false;
        o.clear(i);
        kind[i] = FPX_WIRE_PHASE2B, slot[i] = f.i[2], round[i] = f.i[3];
        if (acceptor_index) acceptor_index[i] = f.i[1];
      } else if (field == 5 && wt == 2) {  // Phase2bNoopRange { group = 1; acceptor = 2; start = 3; end = 4; round = 5 }
        Fields f;
        if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x3e) != 0x3e) return false;
        o.clear(i);
        kind[i] = FPX_WIRE_PHASE2B_NOOP_RANGE, slot[i] = f.i[3], round[i] = f.i[5];
        if (slot_end) slot_end[i] = f.i[4];
        if (group_index) group_index[i] = f.i[1];
        if (acceptor_index) acceptor_index[i] = f.i[2];
      } else {
        r.skip(wt);  // HighWatermark = 1: the leader's business
      }
    }
    return r.ok;
  });
}

int32_t fpx_wire_mencius_decode_acceptor_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                                 int32_t* kind, int32_t* slot, int32_t* slot_end, int32_t* round,
                                                 int32_t* is_noop, int64_t* value_off, int32_t* value_len,
                                                 int32_t* chosen_watermark, int32_t* bad_index) {
  if (n > 0 && (!kind || !slot || !round)) return FPX_EINVAL;
  const MenciusOut o{kind, slot, slot_end, round, is_noop, value_off, value_len, nullptr, nullptr, chosen_watermark, buf};
  return decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    o.clear(i);
    while (r.more()) {
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field == 1 && wt == 2) {  // Phase1a { round = 1; chosen_watermark = 2 }
        Fields f;
        if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x6) != 0x6) return false;
        o.clear(i);
        kind[i] = FPX_WIRE_PHASE1A, round[i] = f.i[1];
        if (chosen_watermark) chosen_watermark[i] = f.i[2];
      } else if (field == 2 && wt == 2) {
        if (!o.with_value(i, r, 3, FPX_WIRE_PHASE2A)) return false;
      } else if (field == 3 && wt == 2) {
        if (!o.range(i, r, true, FPX_WIRE_PHASE2A_NOOP_RANGE)) return false;
      } else {
        r.skip(wt);
      }
    }
    return r.ok;
  });
}

int32_t fpx_wire_mencius_decode_replica_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                                int32_t* kind, int32_t* slot, int32_t* slot_end, int32_t* is_noop,
                                                int64_t* value_off, int32_t* value_len, int32_t* bad_index) {
  if (n > 0 && (!kind || !slot)) return FPX_EINVAL;
  const MenciusOut o{kind, slot, slot_end, nullptr, is_noop, value_off, value_len, nullptr, nullptr, nullptr, buf};
  return decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    o.clear(i);
    while (r.more()) {
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field == 1 && wt == 2) {
        if (!o.with_value(i, r, 2, FPX_WIRE_CHOSEN)) return false;
      } else if (field == 2 && wt == 2) {
        if (!o.range(i, r, false, FPX_WIRE_CHOSEN_NOOP_RANGE)) return false;
      } else {
        r.skip(wt);
      }
    }
    return r.ok;
  });
}

int64_t fpx_wire_mencius_encode_proxy_leader_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round,
                                                     const uint8_t* value, int32_t value_len, int32_t is_noop) {
  return encode_phase2a(out, cap, 2, slot, round, value, value_len, is_noop);
}
int64_t fpx_wire_mencius_encode_acceptor_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round,
                                                 const uint8_t* value, int32_t value_len, int32_t is_noop) {
  return encode_phase2a(out, cap, 2, slot, round, value, value_len, is_noop);
}
static int64_t encode_ints(uint8_t* out, int64_t cap, uint32_t wrapper_field, int k, const int32_t* v) {
  int64_t inner = 0;
  for (int j = 0; j < k; ++j) inner += i32_len(v[j]);
  return wrapped(out, cap, wrapper_field, inner, [&](Writer& w) {
    for (int j = 0; j < k; ++j) w.i32((uint32_t)j + 1, v[j]);
  });
}
int64_t fpx_wire_mencius_encode_proxy_leader_phase2a_noop_range(uint8_t* out, int64_t cap, int32_t slot_start,
                                                                int32_t slot_end, int32_t round) {
  const int32_t v[3] = {slot_start, slot_end, round};
  return encode_ints(out, cap, 3, 3, v);
}
int64_t fpx_wire_mencius_encode_acceptor_phase2a_noop_range(uint8_t* out, int64_t cap, int32_t slot_start,
                                                            int32_t slot_end, int32_t round) {
  const int32_t v[3] = {slot_start, slot_end, round};
  return encode_ints(out, cap, 3, 3, v);
}
int64_t fpx_wire_mencius_encode_acceptor_phase1a(uint8_t* out, int64_t cap, int32_t round, int32_t chosen_watermark) {
  const int32_t v[2] = {round, chosen_watermark};
  return encode_ints(out, cap, 1, 2, v);
}
int64_t fpx_wire_mencius_encode_proxy_leader_phase2b(uint8_t* out, int64_t cap, int32_t acceptor_index, int32_t slot,
                                                     int32_t round) {
  const int32_t v[3] = {acceptor_index, slot, round};
  return encode_ints(out, cap, 4, 3, v);
}
int64_t fpx_wire_mencius_encode_proxy_leader_phase2b_noop_range(uint8_t* out, int64_t cap, int32_t acceptor_group_index,
                                                                int32_t acceptor_index, int32_t slot_start,
                                                                int32_t slot_end, int32_t round) {
  const int32_t v[5] = {acceptor_group_index, acceptor_index, slot_start, slot_end, round};
  return encode_ints(out, cap, 5, 5, v);
}
int64_t fpx_wire_mencius_encode_replica_chosen(uint8_t* out, int64_t cap, int32_t slot, const uint8_t* value,
                                               int32_t value_len, int32_t is_noop) {
  return fpx_wire_encode_replica_chosen(out, cap, slot, value, value_len, is_noop);  // same shape, same field number
}
int64_t fpx_wire_mencius_encode_replica_chosen_noop_range(uint8_t* out, int64_t cap, int32_t slot_start,
                                                          int32_t slot_end) {
  const int32_t v[2] = {slot_start, slot_end};
  return encode_ints(out, cap, 2, 2, v);
}
int64_t fpx_wire_mencius_encode_leader_nack(uint8_t* out, int64_t cap, int32_t round) {
  return encode_ints(out, cap, 7, 1, &round);
}

// ---- EPaxos (epaxos/EPaxos.proto) ------------------------------------------------------------------------------
namespace {

constexpr int64_t CANNOT_ENCODE = -((int64_t)1 << 62);

int64_t pair_len(int32_t a, int32_t b) { return i32_len(a) + i32_len(b); }  // Instance / Ballot body
void put_pair(Writer& w, uint32_t field, int32_t a, int32_t b) {
  w.tag(field, 2);
  w.varint((uint64_t)pair_len(a, b));
  w.i32(1, a);
  w.i32(2, b);
}
int64_t pair_field_len(int32_t a, int32_t b) { return 1 + varint_len((uint64_t)pair_len(a, b)) + pair_len(a, b); }

// InstancePrefixSetProto body
int64_t deps_len(const fpx_wire_epx_msg& m, std::vector<int64_t>* per) {
  int64_t total = i32_len(m.num_replicas);
  per->assign((size_t)m.num_replicas, 0);
  for (int l = 0; l < m.num_replicas; ++l) (*per)[(size_t)l] = i32_len(m.deps_watermark[l]);
  for (int j = 0; j < m.num_values; ++j) (*per)[(size_t)m.values_leader[j]] += i32_len(m.values_id[j]);
  for (int l = 0; l < m.num_replicas; ++l) total += 1 + varint_len((uint64_t)(*per)[(size_t)l]) + (*per)[(size_t)l];
  return total;
}
void put_deps(Writer& w, uint32_t field, const fpx_wire_epx_msg& m, int64_t body, const std::vector<int64_t>& per) {
  w.tag(field, 2);
  w.varint((uint64_t)body);
  w.i32(1, m.num_replicas);
  for (int l = 0; l < m.num_replicas; ++l) {
    w.tag(2, 2);
    w.varint((uint64_t)per[(size_t)l]);
    w.i32(1, m.deps_watermark[l]);
    for (int j = 0; j < m.num_values; ++j)
      if (m.values_leader[j] == l) w.i32(2, m.values_id[j]);  // repeated int32, proto2: one tag per element
  }
}

bool parse_pair(Reader r, int32_t* a, int32_t* b) {
  Fields f;
  if (!parse_flat(r, 0, &f) || (f.seen & 0x6) != 0x6) return false;
  *a = f.i[1], *b = f.i[2];
  return true;
}

struct EpxOne {  // one decoded message
  int32_t kind = FPX_WIRE_OTHER, il = -1, in = -1, bo = -1, br = -1, ri = -1, seq = -1, vbo = -1, vbr = -1, status = -1;
  Value cmd;
  bool has_cmd = false, has_deps = false, has_seq = false;
  int32_t num_replicas = -1;
  std::vector<int32_t> wm;
  std::vector<std::pair<int32_t, int32_t>> values;
};

// IntPrefixSetProto { watermark = 1; repeated int32 value = 2 (unpacked by ScalaPB; packed accepted too) }
bool parse_int_prefix_set(Reader r, int leader, EpxOne* o) {
  bool seen_wm = false;
  int32_t wm = 0;
  while (r.more()) {
    const uint64_t tag = r.varint();
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (field == 1 && wt == 0) {
      wm = as_i32(r.varint());
      seen_wm = true;
    } else if (field == 2 && wt == 0) {
      o->values.push_back({leader, as_i32(r.varint())});
    } else if (field == 2 && wt == 2) {
      Reader p = r.sub();
      while (p.more()) o->values.push_back({leader, as_i32(p.varint())});
      if (!p.ok) return false;
    } else {
      r.skip(wt);
    }
  }
  if (!r.ok || !seen_wm) return false;
  o->wm.push_back(wm);
  return true;
}

bool parse_deps(Reader r, EpxOne* o) {
  bool seen_n = false;
  o->wm.clear();
  o->values.clear();
  int leader = 0;
  while (r.more()) {
    const uint64_t tag = r.varint();
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (field == 1 && wt == 0) {
      o->num_replicas = as_i32(r.varint());
      seen_n = true;
    } else if (field == 2 && wt == 2) {
      Reader sub = r.sub();
      if (!r.ok || !parse_int_prefix_set(sub, leader++, o)) return false;
    } else {
      r.skip(wt);
    }
  }
  // InstancePrefixSet.fromProto (epaxos/InstancePrefixSet.scala:48-53) takes numReplicas and the sets as they come;
  // every producer writes exactly numReplicas of them -- anything else is refused here
  if (!r.ok || !seen_n || o->num_replicas < 0 || (int)o->wm.size() != o->num_replicas) return false;
  o->has_deps = true;
  return true;
}

// the body of one ReplicaInbound member.  layout: which field numbers carry what, per kind
struct EpxLayout {
  int instance, ballot, command, seq, deps, replica, vote_ballot, status;
  unsigned required;  // bits of the field numbers that must be present
};
const EpxLayout* epx_layout(int32_t kind) {
  static const EpxLayout pre_accept{1, 2, 3, 4, 5, 0, 0, 0, 0x3e}, pre_accept_ok{1, 2, 0, 4, 5, 3, 0, 0, 0x3e},
      accept_ok{2, 3, 0, 0, 0, 7, 0, 0, 0x8c}, commit{1, 0, 2, 3, 4, 0, 0, 0, 0x1e}, prepare{1, 2, 0, 0, 0, 0, 0, 0, 0x6},
      prepare_ok{2, 1, 6, 7, 8, 3, 4, 5, 0x3e}, nack{1, 2, 0, 0, 0, 0, 0, 0, 0x6};
  switch (kind) {
    case FPX_WIRE_EPX_PRE_ACCEPT: case FPX_WIRE_EPX_ACCEPT: return &pre_accept;
    case FPX_WIRE_EPX_PRE_ACCEPT_OK: return &pre_accept_ok;
    case FPX_WIRE_EPX_ACCEPT_OK: return &accept_ok;
    case FPX_WIRE_EPX_COMMIT: return &commit;
    case FPX_WIRE_EPX_PREPARE: return &prepare;
    case FPX_WIRE_EPX_PREPARE_OK: return &prepare_ok;
    case FPX_WIRE_EPX_NACK: return &nack;
  }
  return nullptr;
}

bool parse_epx_member(Reader r, int32_t kind, EpxOne* o) {
  const EpxLayout& L = *epx_layout(kind);
  unsigned seen = 0;
  o->kind = kind;
  while (r.more()) {
    const uint64_t tag = r.varint();
    const int field = (int)(tag >> 3);
    const uint32_t wt = (uint32_t)(tag & 7);
    bool known = true;
    if (field == 0) return false;  // no such field (a layout entry of 0 means "this kind has none": never a match)
    if (field == L.instance && wt == 2) {
      if (!parse_pair(r.sub(), &o->il, &o->in)) return false;
    } else if (field == L.ballot && wt == 2) {
      if (!parse_pair(r.sub(), &o->bo, &o->br)) return false;
    } else if (field == L.vote_ballot && wt == 2) {
      if (!parse_pair(r.sub(), &o->vbo, &o->vbr)) return false;
    } else if (field == L.command && wt == 2) {
      Reader sub = r.sub();
      if (!r.ok || !parse_value(sub, &o->cmd)) return false;
      o->has_cmd = true;
    } else if (field == L.deps && wt == 2) {
      Reader sub = r.sub();
      if (!r.ok || !parse_deps(sub, o)) return false;
    } else if (field == L.seq && wt == 0) {
      o->seq = as_i32(r.varint());
      o->has_seq = true;
    } else if (field == L.replica && wt == 0) {
      o->ri = as_i32(r.varint());
    } else if (field == L.status && wt == 0) {
      o->status = as_i32(r.varint());
    } else {
      known = false;
      r.skip(wt);
    }
    if (known && field < 32) seen |= 1u << field;
    if (!r.ok) return false;
  }
  return r.ok && (seen & L.required) == L.required;
}

}  // namespace

int64_t fpx_wire_epaxos_encode_replica_inbound(uint8_t* out, int64_t cap, const fpx_wire_epx_msg* mp) {
  if (!mp) return CANNOT_ENCODE;
  const fpx_wire_epx_msg& m = *mp;
  const EpxLayout* Lp = epx_layout(m.kind);
  if (!Lp) return CANNOT_ENCODE;
  const EpxLayout& L = *Lp;
  const bool is_prepare_ok = m.kind == FPX_WIRE_EPX_PREPARE_OK;
  const bool want_cmd = L.command && (!is_prepare_ok || m.is_noop >= 0);
  const bool want_seq = L.seq && (!is_prepare_ok || m.has_sequence_number);
  const bool want_deps = L.deps && (!is_prepare_ok || m.num_replicas >= 0);
  if (want_deps) {
    if (m.num_replicas < 0 || m.num_replicas > 4096 || (m.num_replicas > 0 && !m.deps_watermark) || m.num_values < 0 ||
        (m.num_values > 0 && (!m.values_leader || !m.values_id)))
      return CANNOT_ENCODE;
    for (int j = 0; j < m.num_values; ++j)
      if (m.values_leader[j] < 0 || m.values_leader[j] >= m.num_replicas) return CANNOT_ENCODE;
  }
  const uint8_t* cmd = m.command;
  int32_t cmd_len = m.command_len;
  if (want_cmd) {
    if (m.is_noop < 0 || (!m.is_noop && (!cmd || cmd_len < 0))) return CANNOT_ENCODE;
    pick_value(cmd, cmd_len, m.is_noop);
  }
  std::vector<int64_t> per;
  const int64_t dlen = want_deps ? deps_len(m, &per) : 0;
  // the fields of the member, in field-number order
  struct Item { int field; int what; };  // what: 0 instance, 1 ballot, 2 command, 3 seq, 4 deps, 5 replica, 6 vote, 7 status
  Item items[8];
  int k = 0;
  auto add = [&](int field, int what, bool present) { if (field && present) items[k++] = Item{field, what}; };
  add(L.instance, 0, true), add(L.ballot, 1, true), add(L.command, 2, want_cmd), add(L.seq, 3, want_seq);
  add(L.deps, 4, want_deps), add(L.replica, 5, true), add(L.vote_ballot, 6, true), add(L.status, 7, true);
  for (int a = 1; a < k; ++a)
    for (int b = a; b > 0 && items[b].field < items[b - 1].field; --b) std::swap(items[b], items[b - 1]);
  int64_t inner = 0;
  for (int a = 0; a < k; ++a) switch (items[a].what) {
      case 0: inner += pair_field_len(m.instance_leader, m.instance_number); break;
      case 1: inner += pair_field_len(m.ballot_ordering, m.ballot_replica); break;
      case 2: inner += 1 + varint_len((uint64_t)cmd_len) + cmd_len; break;
      case 3: inner += i32_len(m.sequence_number); break;
      case 4: inner += 1 + varint_len((uint64_t)dlen) + dlen; break;
      case 5: inner += i32_len(m.replica_index); break;
      case 6: inner += pair_field_len(m.vote_ballot_ordering, m.vote_ballot_replica); break;
      case 7: inner += i32_len(m.status); break;
    }
  const uint32_t wrapper = (uint32_t)(m.kind - 14);  // ReplicaInbound's oneof: pre_accept = 2 ... nack = 9
  return wrapped(out, cap, wrapper, inner, [&](Writer& w) {
    for (int a = 0; a < k; ++a) {
      const uint32_t f = (uint32_t)items[a].field;
      switch (items[a].what) {
        case 0: put_pair(w, f, m.instance_leader, m.instance_number); break;
        case 1: put_pair(w, f, m.ballot_ordering, m.ballot_replica); break;
        case 2: w.tag(f, 2); w.varint((uint64_t)cmd_len); w.bytes(cmd, cmd_len); break;
        case 3: w.i32(f, m.sequence_number); break;
        case 4: put_deps(w, f, m, dlen, per); break;
        case 5: w.i32(f, m.replica_index); break;
        case 6: put_pair(w, f, m.vote_ballot_ordering, m.vote_ballot_replica); break;
        case 7: w.i32(f, m.status); break;
      }
    }
  });
}

int32_t fpx_wire_epaxos_decode_replica_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                               int32_t max_replicas, int32_t* kind, int32_t* instance_leader,
                                               int32_t* instance_number, int32_t* ballot_ordering,
                                               int32_t* ballot_replica, int32_t* replica_index,
                                               int32_t* sequence_number, int32_t* vote_ballot_ordering,
                                               int32_t* vote_ballot_replica, int32_t* status, int32_t* is_noop,
                                               int64_t* cmd_off, int32_t* cmd_len, int32_t* deps_num_replicas,
                                               int32_t* deps_watermark, int64_t* values_off, int64_t values_cap,
                                               int32_t* values_leader, int32_t* values_id, int32_t* bad_index) {
  if (n > 0 && !kind) return FPX_EINVAL;
  if (max_replicas < 0 || values_cap < 0 || (deps_watermark && max_replicas < 1)) return FPX_EINVAL;
  int64_t nvalues = 0;
  if (values_off && n >= 0) values_off[0] = 0;
  EpxOne o;
  const int32_t st = decode_loop(buf, buf_len, offsets, n, bad_index, [&](int32_t i, Reader r) {
    o = EpxOne();
    while (r.more()) {  // the last member of the oneof that is present wins
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field >= 2 && field <= 9 && wt == 2) {
        Reader sub = r.sub();
        o = EpxOne();
        if (!r.ok || !parse_epx_member(sub, (int32_t)field + 14, &o)) return false;
      } else {
        r.skip(wt);  // ClientRequest = 1 goes to the JVM actor
      }
    }
    if (!r.ok) return false;
    kind[i] = o.kind;
    if (instance_leader) instance_leader[i] = o.il;
    if (instance_number) instance_number[i] = o.in;
    if (ballot_ordering) ballot_ordering[i] = o.bo;
    if (ballot_replica) ballot_replica[i] = o.br;
    if (replica_index) replica_index[i] = o.ri;
    if (sequence_number) sequence_number[i] = o.has_seq ? o.seq : -1;
    if (vote_ballot_ordering) vote_ballot_ordering[i] = o.vbo;
    if (vote_ballot_replica) vote_ballot_replica[i] = o.vbr;
    if (status) status[i] = o.status;
    if (is_noop) is_noop[i] = o.has_cmd ? o.cmd.is_noop : -1;
    if (cmd_off) cmd_off[i] = o.has_cmd ? o.cmd.at - buf : -1;
    if (cmd_len) cmd_len[i] = o.has_cmd ? o.cmd.len : -1;
    if (deps_num_replicas) deps_num_replicas[i] = o.has_deps ? o.num_replicas : -1;
    if (deps_watermark) {
      int32_t* row = deps_watermark + (size_t)i * max_replicas;
      for (int l = 0; l < max_replicas; ++l) row[l] = 0;
      if (o.has_deps) {
        if (o.num_replicas > max_replicas) return false;
        for (int l = 0; l < o.num_replicas; ++l) row[l] = o.wm[(size_t)l];
      }
    }
    if (o.has_deps)
      for (const auto& v : o.values) {
        if (nvalues < values_cap) {
          if (values_leader) values_leader[nvalues] = v.first;
          if (values_id) values_id[nvalues] = v.second;
        }
        ++nvalues;
      }
    if (values_off) values_off[i + 1] = nvalues;
    return true;
  });
  if (st != FPX_OK) return st;
  if (nvalues > values_cap && (values_leader || values_id)) return FPX_ECAPACITY;
  return FPX_OK;
}

}  // extern "C"
