// fpx_wire.cpp -- the wire adapter of include/fpx_wire.h: protobuf (proto2, ScalaPB-compatible canonical
// encoding) <-> SoA batches for the messages of the Phase-2 path.  Host code, no dependencies: a varint reader /
// writer and one small parser per message of shared/src/main/scala/frankenpaxos/multipaxos/MultiPaxos.proto.
#include "../../include/fpx_wire.h"

#include <cstring>
#include <unordered_map>

#include "../../include/fpx.h"

namespace {

// ---- reading ---------------------------------------------------------------------------------------------
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;

  bool more() const { return ok && p < end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 70; shift += 7) {
      if (p >= end) break;
      const uint8_t b = *p++;
      if (shift < 64) v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;  // truncated, or longer than 10 bytes
    return 0;
  }
  // a length-delimited field: the sub-range, consumed
  Reader sub() {
    const uint64_t len = varint();
    Reader r{p, p, ok};
    if (!ok || len > (uint64_t)(end - p)) {
      ok = false;
      r.ok = false;
      return r;
    }
    r.end = p + len;
    p += len;
    return r;
  }
  void skip(uint32_t wire_type) {
    switch (wire_type) {
      case 0: (void)varint(); break;
      case 1: if (end - p < 8) ok = false; else p += 8; break;
      case 2: (void)sub(); break;
      case 5: if (end - p < 4) ok = false; else p += 4; break;
      default: ok = false;  // groups are not used by these messages
    }
  }
};

// int32 fields travel as (sign-extended) varints
inline int32_t as_i32(uint64_t v) { return (int32_t)(uint32_t)v; }

struct Value {  // a CommandBatchOrNoop field
  const uint8_t* at = nullptr;
  int32_t len = -1;
  int32_t is_noop = -1;
};

// CommandBatchOrNoop { oneof value { CommandBatch command_batch = 1; Noop noop = 2; } }   MultiPaxos.proto:213-221
bool parse_value(Reader r, Value* out) {
  out->at = r.p;
  out->len = (int32_t)(r.end - r.p);
  int which = 0;
  while (r.more()) {
    const uint64_t tag = r.varint();
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if ((field == 1 || field == 2) && wt == 2) {
      (void)r.sub();
      which = (int)field;  // the last one set wins, as in every protobuf runtime
    } else {
      r.skip(wt);
    }
  }
  if (!r.ok || which == 0) return false;  // logger.fatal("Empty CommandBatchOrNoop") territory: reject
  out->is_noop = which == 2;
  return true;
}

struct Fields {
  int32_t i[5] = {0, 0, 0, 0, 0};  // int32 fields 1..4
  unsigned seen = 0;               // bit f set: field f was present
  Value value;
  bool has_value = false;
};

// a flat message of int32 fields and at most one CommandBatchOrNoop field (number value_field, 0 = none)
bool parse_flat(Reader r, int value_field, Fields* f) {
  while (r.more()) {
    const uint64_t tag = r.varint();
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (value_field && (int)field == value_field && wt == 2) {
      Reader s = r.sub();
      if (!r.ok || !parse_value(s, &f->value)) return false;
      f->has_value = true;
    } else if (field >= 1 && field <= 4 && wt == 0) {
      f->i[field] = as_i32(r.varint());
      f->seen |= 1u << field;
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

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
int32_t decode_loop(const uint8_t* buf, const int64_t* offsets, int32_t n, int32_t* bad_index, Emit emit) {
  if (n < 0 || (n > 0 && (!buf || !offsets))) return FPX_EINVAL;
  for (int32_t i = 0; i < n; ++i) {
    bool ok = offsets[i] >= 0 && offsets[i + 1] >= offsets[i];
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

int32_t fpx_wire_decode_proxy_leader_inbound(const uint8_t* buf, const int64_t* offsets, int32_t n, int32_t* kind,
                                             int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                             int32_t* value_len, int32_t* group_index, int32_t* acceptor_index,
                                             int32_t* bad_index) {
  if (n > 0 && (!kind || !slot || !round)) return FPX_EINVAL;
  return decode_loop(buf, offsets, n, bad_index, [&](int32_t i, Reader r) {
    kind[i] = FPX_WIRE_OTHER, slot[i] = -1, round[i] = -1;
    if (is_noop) is_noop[i] = -1;
    if (value_off) value_off[i] = -1;
    if (value_len) value_len[i] = -1;
    if (group_index) group_index[i] = -1;
    if (acceptor_index) acceptor_index[i] = -1;
    while (r.more()) {  // ProxyLeaderInbound: the last member of the oneof that is present wins
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field == 1 && wt == 2) {  // Phase2a
        Fields f;
        if (!parse_flat(r.sub(), 3, &f) || !r.ok || (f.seen & 0x6) != 0x6 || !f.has_value) return false;
        kind[i] = FPX_WIRE_PHASE2A, slot[i] = f.i[1], round[i] = f.i[2];
        if (is_noop) is_noop[i] = f.value.is_noop;
        if (value_off) value_off[i] = f.value.at - buf;
        if (value_len) value_len[i] = f.value.len;
      } else if (field == 2 && wt == 2) {  // Phase2b
        Fields f;
        if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x1e) != 0x1e) return false;
        kind[i] = FPX_WIRE_PHASE2B, slot[i] = f.i[3], round[i] = f.i[4];
        if (group_index) group_index[i] = f.i[1];
        if (acceptor_index) acceptor_index[i] = f.i[2];
      } else {
        r.skip(wt);
      }
    }
    return r.ok;
  });
}

int32_t fpx_wire_decode_acceptor_inbound(const uint8_t* buf, const int64_t* offsets, int32_t n, int32_t* kind,
                                         int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                         int32_t* value_len, int32_t* chosen_watermark, int32_t* bad_index) {
  if (n > 0 && (!kind || !slot || !round)) return FPX_EINVAL;
  return decode_loop(buf, offsets, n, bad_index, [&](int32_t i, Reader r) {
    kind[i] = FPX_WIRE_OTHER, slot[i] = -1, round[i] = -1;
    if (is_noop) is_noop[i] = -1;
    if (value_off) value_off[i] = -1;
    if (value_len) value_len[i] = -1;
    if (chosen_watermark) chosen_watermark[i] = -1;
    while (r.more()) {
      const uint64_t tag = r.varint();
      const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
      if (field == 1 && wt == 2) {  // Phase1a
        Fields f;
        if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x6) != 0x6) return false;
        kind[i] = FPX_WIRE_PHASE1A, round[i] = f.i[1], slot[i] = -1;
        if (chosen_watermark) chosen_watermark[i] = f.i[2];
      } else if (field == 2 && wt == 2) {  // Phase2a
        Fields f;
        if (!parse_flat(r.sub(), 3, &f) || !r.ok || (f.seen & 0x6) != 0x6 || !f.has_value) return false;
        kind[i] = FPX_WIRE_PHASE2A, slot[i] = f.i[1], round[i] = f.i[2];
        if (is_noop) is_noop[i] = f.value.is_noop;
        if (value_off) value_off[i] = f.value.at - buf;
        if (value_len) value_len[i] = f.value.len;
      } else {
        r.skip(wt);  // MaxSlotRequest, BatchMaxSlotRequest: not this path's
      }
    }
    return r.ok;
  });
}

int32_t fpx_wire_decode_replica_inbound(const uint8_t* buf, const int64_t* offsets, int32_t n, int32_t* kind,
                                        int32_t* slot, int32_t* is_noop, int64_t* value_off, int32_t* value_len,
                                        int32_t* bad_index) {
  if (n > 0 && (!kind || !slot)) return FPX_EINVAL;
  return decode_loop(buf, offsets, n, bad_index, [&](int32_t i, Reader r) {
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
  std::unordered_map<uint64_t, int32_t> index;  // (slot, round) -> row
  index.reserve((size_t)n * 2);
  int32_t rows = 0;
  for (int32_t i = 0; i < n; ++i) {
    if (kind[i] != FPX_WIRE_PHASE2B) continue;
    const int64_t bit = grid_cols > 0 ? (int64_t)group_index[i] * grid_cols + acceptor_index[i] : acceptor_index[i];
    if (bit < 0 || bit >= FPX_MAX_REPLICAS || acceptor_index[i] < 0 || (grid_cols > 0 && acceptor_index[i] >= grid_cols))
      return FPX_EINVAL;
    const uint64_t key = ((uint64_t)(uint32_t)slot[i] << 32) | (uint32_t)round[i];
    auto it = index.find(key);
    int32_t row;
    if (it == index.end()) {
      row = rows++;
      index.emplace(key, row);
      row_slot[row] = slot[i], row_round[row] = round[i];
      memset(row_bits + (size_t)row * 4, 0, 32);
    } else {
      row = it->second;
    }
    row_bits[(size_t)row * 4 + (bit >> 6)] |= 1ull << (bit & 63);
  }
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

}  // extern "C"
