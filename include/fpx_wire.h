/*
 * fpx_wire.h -- wire adapter of libfpx: the reference's protobuf messages of the
 * Phase-2 path <-> the struct-of-arrays batches of include/fpx.h.  Host code only (no GPU involved), plain C ABI.
 *
 * The reference's actors exchange ScalaPB messages serialised with `toByteArray` / parsed with `parseFrom`
 * (shared/src/main/scala/frankenpaxos/ProtoSerializer.scala:8-9) and wrapped in one `...Inbound` oneof per actor:
 *
 *   ProxyLeaderInbound { oneof request { Phase2a phase2a = 1; Phase2b phase2b = 2; } }      MultiPaxos.proto:541-549
 *   AcceptorInbound    { oneof request { Phase1a phase1a = 1; Phase2a phase2a = 2; ... } }   MultiPaxos.proto:551-561
 *   ReplicaInbound     { oneof request { Chosen chosen = 1; ... } }                          MultiPaxos.proto:563-575
 *   LeaderInbound      { oneof request { ...; Nack nack = 6; ... } }                         MultiPaxos.proto:525-539
 *   Phase1a { required int32 round = 1; required int32 chosen_watermark = 2; }               MultiPaxos.proto:238-253
 *   Phase2a { required int32 slot = 1; required int32 round = 2;
 *             required CommandBatchOrNoop command_batch_or_noop = 3; }                       MultiPaxos.proto:273-281
 *   Phase2b { required int32 group_index = 1; required int32 acceptor_index = 2;
 *             required int32 slot = 3; required int32 round = 4; }                           MultiPaxos.proto:283-291
 *   Chosen  { required int32 slot = 1; required CommandBatchOrNoop command_batch_or_noop = 2; }   MultiPaxos.proto:293-299
 *   Nack    { required int32 round = 1; }                                                    MultiPaxos.proto:455-460
 *   CommandBatchOrNoop { oneof value { CommandBatch command_batch = 1; Noop noop = 2; } }    MultiPaxos.proto:213-221
 *
 * A transport wrapper collects the byte arrays one event-loop tick delivers to an actor into ONE
 * buffer plus n + 1 offsets, decodes them here into the SoA arrays of a batch, calls the fpx_* entry point, and
 * encodes the replies (Phase2b / Chosen / Nack) back.  The command payload (CommandBatchOrNoop) never travels to
 * the GPU: the decoder reports where its bytes are, the caller keeps them under the value_id it hands to libfpx
 * (FPX_NOOP for Noop) and splices them back into Chosen when that value_id comes out of the tally.
 *
 * Encoding is canonical protobuf (fields in number order, required fields always present, int32 as varint with
 * negative values sign-extended to 10 bytes), i.e. byte-identical to ScalaPB's toByteArray for these messages;
 * the decoder accepts any valid encoding (fields in any order, unknown fields skipped).
 */
#ifndef FPX_WIRE_H
#define FPX_WIRE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  FPX_WIRE_OTHER = 0,   /* a well-formed message of a kind this path does not handle: left to the JVM actor */
  FPX_WIRE_PHASE2A = 1,
  FPX_WIRE_PHASE2B = 2,
  FPX_WIRE_PHASE1A = 3,
  FPX_WIRE_CHOSEN = 4,
  FPX_WIRE_NACK = 5
};

/* Decodes n ProxyLeaderInbound messages: message i is buf[offsets[i] .. offsets[i + 1]).  Per message:
 * kind[i] (PHASE2A / PHASE2B / OTHER); Phase2a: slot, round, is_noop, and the location of the serialised
 * CommandBatchOrNoop inside buf (value_off, value_len); Phase2b: group_index, acceptor_index, slot, round.
 * Fields that do not apply are set to -1.  Returns FPX_OK, or FPX_EINVAL at the first malformed message (truncated
 * varint, length past the end, missing required field); *bad_index (may be NULL) tells which. */
int32_t fpx_wire_decode_proxy_leader_inbound(const uint8_t* buf, const int64_t* offsets, int32_t n, int32_t* kind,
                                             int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                             int32_t* value_len, int32_t* group_index, int32_t* acceptor_index,
                                             int32_t* bad_index);
/* Decodes n AcceptorInbound messages: kind PHASE1A (round, chosen_watermark) / PHASE2A (slot, round, value) /
 * OTHER (MaxSlotRequest ...). */
int32_t fpx_wire_decode_acceptor_inbound(const uint8_t* buf, const int64_t* offsets, int32_t n, int32_t* kind,
                                         int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                         int32_t* value_len, int32_t* chosen_watermark, int32_t* bad_index);
/* Decodes n ReplicaInbound messages: kind CHOSEN (slot, value) / OTHER. */
int32_t fpx_wire_decode_replica_inbound(const uint8_t* buf, const int64_t* offsets, int32_t n, int32_t* kind,
                                        int32_t* slot, int32_t* is_noop, int64_t* value_off, int32_t* value_len,
                                        int32_t* bad_index);

/* Folds decoded Phase2b messages into the rows fpx_proxy_phase2b takes: one row per distinct (slot, round), in
 * order of first appearance, with the acceptors that answered as a 256-bit set.  Bit of a message =
 * acceptor_index when grid_cols == 0 (non-flexible: the acceptor group follows from the slot,
 * multipaxos/ProxyLeader.scala:190) or group_index * grid_cols + acceptor_index for a grid (group_index = row,
 * Grid.scala).  Messages whose kind is not PHASE2B are skipped.  row_* have room for n rows; *num_rows is set.
 * FPX_EINVAL if a bit falls outside 0..255. */
int32_t fpx_wire_phase2b_rows(int32_t n, const int32_t* kind, const int32_t* group_index,
                              const int32_t* acceptor_index, const int32_t* slot, const int32_t* round,
                              int32_t grid_cols, int32_t* num_rows, int32_t* row_slot, int32_t* row_round,
                              uint64_t* row_bits /* n x 4 */);

/* Encoders.  Each writes ONE wrapped message to out and returns its length, or the negated length needed when
 * cap is too small (nothing written).  value / value_len: the serialised CommandBatchOrNoop as the decoder
 * located it; is_noop != 0 encodes CommandBatchOrNoop{noop} and ignores value. */
int64_t fpx_wire_encode_proxy_leader_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round,
                                             const uint8_t* value, int32_t value_len, int32_t is_noop);
int64_t fpx_wire_encode_acceptor_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round,
                                         const uint8_t* value, int32_t value_len, int32_t is_noop);
int64_t fpx_wire_encode_acceptor_phase1a(uint8_t* out, int64_t cap, int32_t round, int32_t chosen_watermark);
int64_t fpx_wire_encode_proxy_leader_phase2b(uint8_t* out, int64_t cap, int32_t group_index, int32_t acceptor_index,
                                             int32_t slot, int32_t round);
int64_t fpx_wire_encode_replica_chosen(uint8_t* out, int64_t cap, int32_t slot, const uint8_t* value,
                                       int32_t value_len, int32_t is_noop);
int64_t fpx_wire_encode_leader_nack(uint8_t* out, int64_t cap, int32_t round);

/* The replies of one K1 batch as wire bytes: for message i every acceptor in vote_bits[i] answers
 * ProxyLeaderInbound{Phase2b(group_index, acceptor_index, slot[i], round[i])} (Acceptor.scala:211-219).  The bit
 * -> (group_index, acceptor_index) map is the inverse of fpx_wire_phase2b_rows (group_of_slot[i] supplies the
 * group when grid_cols == 0; may be NULL for group 0).  Messages are written back to back into out;
 * out_offsets gets count + 1 entries (capacity max_msgs + 1).  Returns the number of messages, or -1 if out or
 * out_offsets is too small. */
int64_t fpx_wire_encode_phase2b_batch(int32_t n, const int32_t* slot, const int32_t* round,
                                      const uint64_t* vote_bits, const int32_t* group_of_slot, int32_t grid_cols,
                                      uint8_t* out, int64_t cap, int64_t* out_offsets, int64_t max_msgs);

#ifdef __cplusplus
}
#endif
#endif /* FPX_WIRE_H */
