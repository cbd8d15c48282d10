// fpx_wire_parse.hpp -- the protobuf READER of the wire adapter (include/fpx_wire.h), shared by the host decoders of
// fpx_wire.cpp (g++) and the device decoder k_wire_decode_pli of fpx_api.hip (hipcc): one source for both, so a tick of
// messages decodes to the same fields byte for byte whichever side parses it.  Proto2 wire format as ScalaPB writes and
// every protobuf runtime reads it: any field order, unknown fields skipped, the last member of a oneof wins.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FPX_HD __host__ __device__
#else
#define FPX_HD
#endif

namespace fpxw {

// ---- reading ---------------------------------------------------------------------------------------------
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;

  FPX_HD bool more() const { return ok && p < end; }
  FPX_HD uint64_t varint() {
    if (p < end && !(*p & 0x80)) return *p++;  // tags and small values: one byte
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
  FPX_HD Reader sub() {
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
  FPX_HD void skip(uint32_t wire_type) {
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
FPX_HD inline int32_t as_i32(uint64_t v) { return (int32_t)(uint32_t)v; }

struct Value {  // a CommandBatchOrNoop field
  const uint8_t* at = nullptr;
  int32_t len = -1;
  int32_t is_noop = -1;
};

// CommandBatchOrNoop { oneof value { CommandBatch command_batch = 1; Noop noop = 2; } }   MultiPaxos.proto:213-221
FPX_HD inline bool parse_value(Reader r, Value* out) {
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
  int32_t i[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // int32 fields 1..7
  unsigned seen = 0;               // bit f set: field f was present
  Value value;
  bool has_value = false;
};

// a flat message of int32 fields and at most one CommandBatchOrNoop field (number value_field, 0 = none)
FPX_HD inline bool parse_flat(Reader r, int value_field, Fields* f) {
  while (r.more()) {
    const uint64_t tag = r.varint();
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (value_field && (int)field == value_field && wt == 2) {
      Reader s = r.sub();
      if (!r.ok || !parse_value(s, &f->value)) return false;
      f->has_value = true;
    } else if (field >= 1 && field <= 7 && wt == 0) {
      f->i[field] = as_i32(r.varint());
      f->seen |= 1u << field;
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

// ---- whole messages of multipaxos/MultiPaxos.proto, one output record per message -------------------------
// kind codes as in include/fpx_wire.h (FPX_WIRE_*); fields that do not apply stay -1
struct Msg {
  int32_t kind = 0, slot = -1, round = -1, is_noop = -1, value_len = -1;
  int32_t a = -1;  // ProxyLeaderInbound: Phase2b.group_index     AcceptorInbound: Phase1a.chosen_watermark
  int32_t b = -1;  // ProxyLeaderInbound: Phase2b.acceptor_index
  int64_t value_off = -1;  // of the serialised CommandBatchOrNoop, from `base`
};

// ProxyLeaderInbound { oneof request { Phase2a phase2a = 1; Phase2b phase2b = 2; } }   MultiPaxos.proto:286-291
FPX_HD inline bool parse_proxy_leader_inbound(const uint8_t* base, Reader r, Msg* o) {
  while (r.more()) {  // the last member of the oneof that is present wins
    const uint64_t tag = r.varint();
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (field == 1 && wt == 2) {  // Phase2a { slot = 1; round = 2; command_batch_or_noop = 3 }
      Fields f;
      if (!parse_flat(r.sub(), 3, &f) || !r.ok || (f.seen & 0x6) != 0x6 || !f.has_value) return false;
      *o = Msg();
      o->kind = 1, o->slot = f.i[1], o->round = f.i[2];
      o->is_noop = f.value.is_noop, o->value_off = f.value.at - base, o->value_len = f.value.len;
    } else if (field == 2 && wt == 2) {  // Phase2b { group_index = 1; acceptor_index = 2; slot = 3; round = 4 }
      Fields f;
      if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x1e) != 0x1e) return false;
      *o = Msg();
      o->kind = 2, o->slot = f.i[3], o->round = f.i[4], o->a = f.i[1], o->b = f.i[2];
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

// AcceptorInbound { oneof request { Phase1a phase1a = 1; Phase2a phase2a = 2; MaxSlotRequest max_slot_request = 3;
//                                   BatchMaxSlotRequest batch_max_slot_request = 4; } }   MultiPaxos.proto:550-562
FPX_HD inline bool parse_acceptor_inbound(const uint8_t* base, Reader r, Msg* o) {
  while (r.more()) {
    const uint64_t tag = r.varint();
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (field == 1 && wt == 2) {  // Phase1a { round = 1; chosen_watermark = 2 }
      Fields f;
      if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x6) != 0x6) return false;
      *o = Msg();
      o->kind = 3, o->round = f.i[1], o->a = f.i[2];
    } else if (field == 2 && wt == 2) {  // Phase2a
      Fields f;
      if (!parse_flat(r.sub(), 3, &f) || !r.ok || (f.seen & 0x6) != 0x6 || !f.has_value) return false;
      *o = Msg();
      o->kind = 1, o->slot = f.i[1], o->round = f.i[2];
      o->is_noop = f.value.is_noop, o->value_off = f.value.at - base, o->value_len = f.value.len;
    } else if (field == 3 && wt == 2) {  // MaxSlotRequest { CommandId command_id = 1 }   MultiPaxos.proto:316-321
      // (the linearizable read path, Acceptor.scala:222-237: the reply carries the CommandId back unchanged, so what the
      // caller gets is WHERE the serialised CommandId lies -- value_off / value_len -- after a check of its required fields)
      Reader m = r.sub();
      if (!r.ok) return false;
      bool have = false;
      const uint8_t* at = nullptr;
      int32_t len = -1;
      while (m.more()) {
        const uint64_t t2 = m.varint();
        const uint32_t f2 = (uint32_t)(t2 >> 3), w2 = (uint32_t)(t2 & 7);
        if (f2 == 1 && w2 == 2) {  // CommandId { bytes client_address = 1; int32 client_pseudonym = 2; int32 client_id = 3 }
          Reader c = m.sub();
          if (!m.ok) return false;
          at = c.p, len = (int32_t)(c.end - c.p);
          unsigned seen = 0;
          while (c.more()) {
            const uint64_t t3 = c.varint();
            const uint32_t f3 = (uint32_t)(t3 >> 3), w3 = (uint32_t)(t3 & 7);
            if (f3 == 1 && w3 == 2) (void)c.sub(), seen |= 2u;
            else if ((f3 == 2 || f3 == 3) && w3 == 0) (void)c.varint(), seen |= 1u << f3;
            else c.skip(w3);
          }
          if (!c.ok || seen != 0xeu) return false;
          have = true;
        } else {
          m.skip(w2);
        }
      }
      if (!m.ok || !have) return false;
      *o = Msg();
      o->kind = 10, o->value_off = at - base, o->value_len = len;
    } else if (field == 4 && wt == 2) {  // BatchMaxSlotRequest { read_batcher_index = 1; read_batcher_id = 2 }   :333-339
      Fields f;
      if (!parse_flat(r.sub(), 0, &f) || !r.ok || (f.seen & 0x6) != 0x6) return false;
      *o = Msg();
      o->kind = 11, o->slot = f.i[1], o->round = f.i[2];  // slot = read_batcher_index, round = read_batcher_id
    } else {
      r.skip(wt);
    }
  }
  return r.ok;
}

// message i of a tick is buf[offsets[i] .. offsets[i + 1]): is offsets[i] a legal boundary?
FPX_HD inline bool offset_ok(const int64_t* offsets, int32_t i, int64_t buf_len) {
  return offsets[i] >= 0 && offsets[i] <= buf_len && (i == 0 || offsets[i] >= offsets[i - 1]);
}

}  // namespace fpxw
