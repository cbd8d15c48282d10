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
  FPX_WIRE_NACK = 5,
  FPX_WIRE_PHASE1B = 9,
  /* the acceptor's read path (multipaxos/Acceptor.scala:222-254) */
  FPX_WIRE_MAX_SLOT_REQUEST = 10,
  FPX_WIRE_BATCH_MAX_SLOT_REQUEST = 11,
  /* mencius/Mencius.proto */
  FPX_WIRE_PHASE2A_NOOP_RANGE = 6,
  FPX_WIRE_PHASE2B_NOOP_RANGE = 7,
  FPX_WIRE_CHOSEN_NOOP_RANGE = 8,
  /* epaxos/EPaxos.proto: the members of ReplicaInbound, by their field number + 14 */
  FPX_WIRE_EPX_PRE_ACCEPT = 16,
  FPX_WIRE_EPX_PRE_ACCEPT_OK = 17,
  FPX_WIRE_EPX_ACCEPT = 18,
  FPX_WIRE_EPX_ACCEPT_OK = 19,
  FPX_WIRE_EPX_COMMIT = 20,
  FPX_WIRE_EPX_PREPARE = 21,
  FPX_WIRE_EPX_PREPARE_OK = 22,
  FPX_WIRE_EPX_NACK = 23
};

/* Decodes n ProxyLeaderInbound messages: message i is buf[offsets[i] .. offsets[i + 1]); buf_len = the bytes buf
 * holds: the n + 1 offsets must be non-negative, non-decreasing and at most buf_len, which is checked for ALL of them
 * before anything is parsed (FPX_EINVAL, *bad_index = the first offender).  Per message:
 * kind[i] (PHASE2A / PHASE2B / OTHER); Phase2a: slot, round, is_noop, and the location of the serialised
 * CommandBatchOrNoop inside buf (value_off, value_len); Phase2b: group_index, acceptor_index, slot, round.
 * Fields that do not apply are set to -1.  Returns FPX_OK, or FPX_EINVAL at the first malformed message (truncated
 * varint, length past the end, missing required field); *bad_index (may be NULL) tells which. */
int32_t fpx_wire_decode_proxy_leader_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n, int32_t* kind,
                                             int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                             int32_t* value_len, int32_t* group_index, int32_t* acceptor_index,
                                             int32_t* bad_index);
/* Decodes n AcceptorInbound messages: kind PHASE1A (round, chosen_watermark) / PHASE2A (slot, round, value) /
 * MAX_SLOT_REQUEST (value_off, value_len = where the serialised CommandId lies: the reply returns it unchanged) /
 * BATCH_MAX_SLOT_REQUEST (slot = read_batcher_index, round = read_batcher_id) / OTHER. */
int32_t fpx_wire_decode_acceptor_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n, int32_t* kind,
                                         int32_t* slot, int32_t* round, int32_t* is_noop, int64_t* value_off,
                                         int32_t* value_len, int32_t* chosen_watermark, int32_t* bad_index);
/* Decodes n ReplicaInbound messages: kind CHOSEN (slot, value) / OTHER. */
int32_t fpx_wire_decode_replica_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n, int32_t* kind,
                                        int32_t* slot, int32_t* is_noop, int64_t* value_off, int32_t* value_len,
                                        int32_t* bad_index);

/* The two decoders above ON THE DEVICE: the tick's bytes and its n + 1 offsets are device pointers (or page-locked host
 * memory the GPU can read), the outputs are device arrays of n elements, and the work is enqueued on the context's
 * stream like every _dev entry point of include/fpx.h -- a tick goes  copy -> decode -> fpx_phase2_fused_dev /
 * fpx_acceptor_phase2a_dev  without the host parsing a byte.  One thread per message runs the SAME parser the host
 * decoders run (csrc/fpx_wire_parse.hpp is compiled for both sides), so the fields are identical: kind, slot, round,
 * is_noop, value_off, value_len and group_index / acceptor_index (resp. chosen_watermark), -1 where a field does not
 * apply; d_kind, d_slot and d_round are required, the others may be NULL.  d_value_id (may be NULL): value_id_base + i
 * for a Phase2a, -1 otherwise -- the id under which the caller files message i's command bytes (value_off, value_len)
 * and which the Phase-2 kernels carry through to Chosen.
 * Errors follow the _dev convention: the call returns FPX_OK once enqueued; a bad offset or a malformed message makes
 * the context's status FPX_EINVAL (fpx_sync; fpx_error_detail's index = the first bad offset, else the first
 * malformed message, as *bad_index above) and every later _dev call up to that fpx_sync applies nothing, so a
 * half-decoded tick never reaches the acceptors.  The outputs of a failed decode are unspecified.  n < 2^30. */
struct fpx_ctx;
int32_t fpx_wire_decode_proxy_leader_inbound_dev(struct fpx_ctx* ctx, const uint8_t* d_buf, int64_t buf_len,
                                                 const int64_t* d_offsets, int32_t n, int32_t* d_kind, int32_t* d_slot,
                                                 int32_t* d_round, int32_t* d_is_noop, int64_t* d_value_off,
                                                 int32_t* d_value_len, int32_t* d_group_index,
                                                 int32_t* d_acceptor_index, int32_t value_id_base, int32_t* d_value_id);
int32_t fpx_wire_decode_acceptor_inbound_dev(struct fpx_ctx* ctx, const uint8_t* d_buf, int64_t buf_len,
                                             const int64_t* d_offsets, int32_t n, int32_t* d_kind, int32_t* d_slot,
                                             int32_t* d_round, int32_t* d_is_noop, int64_t* d_value_off,
                                             int32_t* d_value_len, int32_t* d_chosen_watermark, int32_t value_id_base,
                                             int32_t* d_value_id);

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
/* The acceptor's answers on the read path (multipaxos/Acceptor.scala:222-254):
 *   ClientInbound { oneof request { ...; MaxSlotReply max_slot_reply = 4; ... } }                 MultiPaxos.proto:489-499
 *   MaxSlotReply { CommandId command_id = 1; group_index = 2; acceptor_index = 3; slot = 4 }      :323-331
 *   ReadBatcherInbound { oneof request { ...; BatchMaxSlotReply batch_max_slot_reply = 4; } }     :513-523
 *   BatchMaxSlotReply { read_batcher_index = 1; read_batcher_id = 2; acceptor_index = 3; slot = 4 }   :341-349
 * command_id / command_id_len: the serialised CommandId as the decoder located it.  slot = Acceptor.maxVotedSlot (-1 before
 * the first vote: a negative int32 is a ten-byte varint). */
int64_t fpx_wire_encode_client_max_slot_reply(uint8_t* out, int64_t cap, const uint8_t* command_id, int32_t command_id_len,
                                              int32_t group_index, int32_t acceptor_index, int32_t slot);
int64_t fpx_wire_encode_read_batcher_batch_max_slot_reply(uint8_t* out, int64_t cap, int32_t read_batcher_index,
                                                          int32_t read_batcher_id, int32_t acceptor_index, int32_t slot);

/* The acceptor's answer to a Phase1a (multipaxos/Acceptor.scala:163-181), as the Leader parses it:
 *   LeaderInbound { oneof request { Phase1b phase1b = 1; ...; Nack nack = 6; ... } }          MultiPaxos.proto:525-539
 *   Phase1b { required int32 group_index = 1; required int32 acceptor_index = 2; required int32 round = 3;
 *             repeated Phase1bSlotInfo info = 4; }                                            MultiPaxos.proto:263-271
 *   Phase1bSlotInfo { required int32 slot = 1; required int32 vote_round = 2;
 *                     required CommandBatchOrNoop vote_value = 3; }                           MultiPaxos.proto:254-261
 * Entry j's CommandBatchOrNoop body is values[value_off[j] .. + value_len[j]) (the JVM-side bytes kept under the
 * value id fpx_acceptor_phase1b_info returned), or Noop where is_noop[j] != 0 (is_noop may be NULL: none is). */
int64_t fpx_wire_encode_leader_phase1b(uint8_t* out, int64_t cap, int32_t group_index, int32_t acceptor_index,
                                       int32_t round, int32_t n_info, const int32_t* slot, const int32_t* vote_round,
                                       const uint8_t* values, const int64_t* value_off, const int32_t* value_len,
                                       const uint8_t* is_noop);
/* Decodes n LeaderInbound messages: kind PHASE1B (round, group_index, acceptor_index, info) / NACK (round) / OTHER.
 * The Phase1bSlotInfo entries of all messages go, in order, into the info_* arrays (capacity info_cap entries;
 * FPX_EINVAL if there are more): message i owns entries info_first[i] .. info_first[i] + info_count[i]. */
int32_t fpx_wire_decode_leader_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                       int32_t* kind, int32_t* round, int32_t* group_index, int32_t* acceptor_index,
                                       int32_t* info_first, int32_t* info_count, int32_t info_cap, int32_t* info_total,
                                       int32_t* info_slot, int32_t* info_vote_round, int32_t* info_is_noop,
                                       int64_t* info_value_off, int32_t* info_value_len, int32_t* bad_index);

/* The replies of one K1 batch as wire bytes: for message i every acceptor in vote_bits[i] answers
 * ProxyLeaderInbound{Phase2b(group_index, acceptor_index, slot[i], round[i])} (Acceptor.scala:211-219).  The bit
 * -> (group_index, acceptor_index) map is the inverse of fpx_wire_phase2b_rows (group_of_slot[i] supplies the
 * group when grid_cols == 0; may be NULL for group 0).  Messages are written back to back into out;
 * out_offsets gets count + 1 entries (capacity max_msgs + 1).  Returns the number of messages, or -1 if out or
 * out_offsets is too small. */
int64_t fpx_wire_encode_phase2b_batch(int32_t n, const int32_t* slot, const int32_t* round,
                                      const uint64_t* vote_bits, const int32_t* group_of_slot, int32_t grid_cols,
                                      uint8_t* out, int64_t cap, int64_t* out_offsets, int64_t max_msgs);

/* ---- Mencius (shared/src/main/scala/frankenpaxos/mencius/Mencius.proto) ---------------------------------------
 *
 *   ProxyLeaderInbound { oneof request { HighWatermark high_watermark = 1; Phase2a phase2a = 2;
 *                        Phase2aNoopRange phase2a_noop_range = 3; Phase2b phase2b = 4;
 *                        Phase2bNoopRange phase2b_noop_range = 5; } }                         Mencius.proto:339-350
 *   AcceptorInbound    { oneof request { Phase1a phase1a = 1; Phase2a phase2a = 2;
 *                        Phase2aNoopRange phase2a_noop_range = 3; } }                         Mencius.proto:352-361
 *   ReplicaInbound     { oneof request { Chosen chosen = 1; ChosenNoopRange chosen_noop_range = 2; } }   :363-371
 *   LeaderInbound      { oneof request { ...; Nack nack = 7; ... } }                          Mencius.proto:322-337
 *   Phase2a          { slot = 1; round = 2; command_batch_or_noop = 3; }                      Mencius.proto:151-158
 *   Phase2aNoopRange { slot_start_inclusive = 1; slot_end_exclusive = 2; round = 3; }         Mencius.proto:160-168
 *   Phase2b          { acceptor_index = 1; slot = 2; round = 3; }   (no group: it follows from the slot)  :169-176
 *   Phase2bNoopRange { acceptor_group_index = 1; acceptor_index = 2; slot_start_inclusive = 3;
 *                      slot_end_exclusive = 4; round = 5; }                                   Mencius.proto:178-187
 *   Chosen           { slot = 1; command_batch_or_noop = 2; }                                 Mencius.proto:189-195
 *   ChosenNoopRange  { slot_start_inclusive = 1; slot_end_exclusive = 2; }                    Mencius.proto:197-203
 *   Phase1a { round = 1; chosen_watermark = 2; }   Nack { round = 1; }                        :104-117, 266-271
 *
 * Decoders: as above (buf, buf_len, offsets, n); slot = the slot, or slot_start_inclusive of a range; slot_end =
 * slot_end_exclusive of a range (-1 otherwise); fields that do not apply are -1; any output array but kind, slot and
 * round may be NULL. */
int32_t fpx_wire_mencius_decode_proxy_leader_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets,
                                                     int32_t n, int32_t* kind, int32_t* slot, int32_t* slot_end,
                                                     int32_t* round, int32_t* is_noop, int64_t* value_off,
                                                     int32_t* value_len, int32_t* group_index, int32_t* acceptor_index,
                                                     int32_t* bad_index);
int32_t fpx_wire_mencius_decode_acceptor_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                                 int32_t* kind, int32_t* slot, int32_t* slot_end, int32_t* round,
                                                 int32_t* is_noop, int64_t* value_off, int32_t* value_len,
                                                 int32_t* chosen_watermark, int32_t* bad_index);
/* kind CHOSEN / CHOSEN_NOOP_RANGE / OTHER; round is not part of these messages (the array is not taken) */
int32_t fpx_wire_mencius_decode_replica_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                                int32_t* kind, int32_t* slot, int32_t* slot_end, int32_t* is_noop,
                                                int64_t* value_off, int32_t* value_len, int32_t* bad_index);
/* Encoders: one wrapped message each, return convention as above. */
int64_t fpx_wire_mencius_encode_proxy_leader_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round,
                                                     const uint8_t* value, int32_t value_len, int32_t is_noop);
int64_t fpx_wire_mencius_encode_acceptor_phase2a(uint8_t* out, int64_t cap, int32_t slot, int32_t round,
                                                 const uint8_t* value, int32_t value_len, int32_t is_noop);
int64_t fpx_wire_mencius_encode_proxy_leader_phase2a_noop_range(uint8_t* out, int64_t cap, int32_t slot_start,
                                                                int32_t slot_end, int32_t round);
int64_t fpx_wire_mencius_encode_acceptor_phase2a_noop_range(uint8_t* out, int64_t cap, int32_t slot_start,
                                                            int32_t slot_end, int32_t round);
int64_t fpx_wire_mencius_encode_acceptor_phase1a(uint8_t* out, int64_t cap, int32_t round, int32_t chosen_watermark);
int64_t fpx_wire_mencius_encode_proxy_leader_phase2b(uint8_t* out, int64_t cap, int32_t acceptor_index, int32_t slot,
                                                     int32_t round);
int64_t fpx_wire_mencius_encode_proxy_leader_phase2b_noop_range(uint8_t* out, int64_t cap, int32_t acceptor_group_index,
                                                                int32_t acceptor_index, int32_t slot_start,
                                                                int32_t slot_end, int32_t round);
int64_t fpx_wire_mencius_encode_replica_chosen(uint8_t* out, int64_t cap, int32_t slot, const uint8_t* value,
                                               int32_t value_len, int32_t is_noop);
int64_t fpx_wire_mencius_encode_replica_chosen_noop_range(uint8_t* out, int64_t cap, int32_t slot_start,
                                                          int32_t slot_end);
int64_t fpx_wire_mencius_encode_leader_nack(uint8_t* out, int64_t cap, int32_t round);

/* ---- EPaxos (shared/src/main/scala/frankenpaxos/epaxos/EPaxos.proto) -------------------------------------------
 *
 *   ReplicaInbound { oneof request { ClientRequest client_request = 1; PreAccept pre_accept = 2;
 *                    PreAcceptOk pre_accept_ok = 3; Accept accept = 4; AcceptOk accept_ok = 5; Commit commit = 6;
 *                    Prepare prepare = 7; PrepareOk prepare_ok = 8; Nack nack = 9; } }          EPaxos.proto:220-235
 *   Instance { replica_index = 1; instance_number = 2; }   Ballot { ordering = 1; replica_index = 2; }   :35-53
 *   CommandOrNoop { oneof value { Command command = 1; Noop noop = 2; } }                       EPaxos.proto:80-89
 *   InstancePrefixSetProto { numReplicas = 1; repeated IntPrefixSetProto int_prefix_set = 2; }  EPaxos.proto:97-104
 *   IntPrefixSetProto { watermark = 1; repeated int32 value = 2; }               compact/IntPrefixSet.proto:12-15
 *   PreAccept   { instance = 1; ballot = 2; command_or_noop = 3; sequence_number = 4; dependencies = 5; }   :113-123
 *   PreAcceptOk { instance = 1; ballot = 2; replica_index = 3; sequence_number = 4; dependencies = 5; }     :125-135
 *   Accept      { instance = 1; ballot = 2; command_or_noop = 3; sequence_number = 4; dependencies = 5; }   :137-147
 *   AcceptOk    { instance = 2; ballot = 3; replica_index = 7; }                                            :149-156
 *   Commit      { instance = 1; command_or_noop = 2; sequence_number = 3; dependencies = 4; }               :158-166
 *   Prepare     { instance = 1; ballot = 2; }                                                               :178-184
 *   PrepareOk   { ballot = 1; instance = 2; replica_index = 3; vote_ballot = 4; status = 5 (NotSeen 0 /
 *                 PreAccepted 1 / Accepted 2); optional command_or_noop = 6, sequence_number = 7,
 *                 dependencies = 8; }                                                                       :186-209
 *   Nack        { instance = 1; largest_ballot = 2; }                                                       :211-218
 *
 * One message as plain fields.  What a kind does not carry is -1 (is_noop: -1 = no CommandOrNoop present,
 * num_replicas: -1 = no dependencies present).  Dependencies: deps_watermark[l] for l < num_replicas and the
 * explicit ids (values_leader[j], values_id[j]), j < num_values.  ballot_* is the message's ballot (Nack:
 * largest_ballot). */
typedef struct {
  int32_t kind; /* FPX_WIRE_EPX_* */
  int32_t instance_leader, instance_number;
  int32_t ballot_ordering, ballot_replica;
  int32_t replica_index;                            /* PreAcceptOk, AcceptOk, PrepareOk */
  int32_t sequence_number;                          /* -1 with has_sequence_number = 0 */
  int32_t has_sequence_number;
  int32_t vote_ballot_ordering, vote_ballot_replica, status; /* PrepareOk */
  int32_t is_noop;                                  /* 1 Noop, 0 command, -1 none */
  const uint8_t* command;                           /* the serialised CommandOrNoop (as the decoder locates it) */
  int32_t command_len;
  int32_t num_replicas;                             /* -1: no dependencies */
  const int32_t* deps_watermark;
  int32_t num_values;
  const int32_t* values_leader;
  const int32_t* values_id;
} fpx_wire_epx_msg;

/* Encodes ONE ReplicaInbound (return convention as above; -(1 << 62) for a message that cannot be encoded: unknown
 * kind, a required part missing).  Explicit ids are written in the order given (ScalaPB writes `values.toSeq` of a
 * hash set: no canonical order exists; any order decodes to the same set). */
int64_t fpx_wire_epaxos_encode_replica_inbound(uint8_t* out, int64_t cap, const fpx_wire_epx_msg* msg);

/* Decodes n ReplicaInbound messages into SoA (every array has n entries unless said otherwise; any but kind may be
 * NULL).  deps_watermark is n x max_replicas (row i holds deps_num_replicas[i] <= max_replicas watermarks, the rest
 * 0; more replicas than max_replicas is FPX_EINVAL at that message).  Explicit ids of all messages back to back:
 * values of message i are values_leader / values_id [values_off[i] .. values_off[i + 1]) (values_off has n + 1
 * entries); if they do not fit values_cap the call returns FPX_ECAPACITY with values_off[n] = the capacity needed
 * (everything else is filled in) -- call again.  cmd_off / cmd_len locate the serialised CommandOrNoop in buf. */
int32_t fpx_wire_epaxos_decode_replica_inbound(const uint8_t* buf, int64_t buf_len, const int64_t* offsets, int32_t n,
                                               int32_t max_replicas, int32_t* kind, int32_t* instance_leader,
                                               int32_t* instance_number, int32_t* ballot_ordering,
                                               int32_t* ballot_replica, int32_t* replica_index,
                                               int32_t* sequence_number, int32_t* vote_ballot_ordering,
                                               int32_t* vote_ballot_replica, int32_t* status, int32_t* is_noop,
                                               int64_t* cmd_off, int32_t* cmd_len, int32_t* deps_num_replicas,
                                               int32_t* deps_watermark, int64_t* values_off, int64_t values_cap,
                                               int32_t* values_leader, int32_t* values_id, int32_t* bad_index);

#ifdef __cplusplus
}
#endif
#endif /* FPX_WIRE_H */
