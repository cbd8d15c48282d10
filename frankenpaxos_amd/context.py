"""Context: a Python handle on one libfpx context (= the acceptor groups + one proxy leader of
SURVEY.md section 8 living in the HBM of one MI355X).

Host-pointer methods take / return numpy arrays and accept any batch; `*_dev` methods take torch
CUDA tensors (already resident in HBM), enqueue on the context's stream and return immediately.
Every method ends up in a HIP kernel; nothing here computes protocol results on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FpxConfig, FpxError


def make_config(num_slots, num_replicas, num_groups=1, num_leader_groups=1, f=0,
                quorum_kind=_lib.FPX_Q_THRESHOLD, grid_rows=0, grid_cols=0, num_leaders=2,
                ballot_mode=_lib.FPX_BALLOT_ACCEPTOR, tally_ways=4, replica_base=0, replicas_total=0,
                device=0, flags=0):
    return FpxConfig(num_slots, num_replicas, num_groups, num_leader_groups, f, quorum_kind,
                     grid_rows, grid_cols, num_leaders, ballot_mode, tally_ways, replica_base,
                     replicas_total, device, flags)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _u64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint64)


def _hp(a):
    return None if a is None else a.ctypes.data


def _dp(t):
    """device pointer of a torch tensor (or None)"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device entry points take contiguous CUDA tensors"
    return t.data_ptr()


class Context:
    def __init__(self, cfg):
        self.L = _lib.lib()
        self.cfg = cfg
        st = self.L.fpx_config_check(C.byref(cfg))
        if st:
            raise FpxError(st, "fpx_config_check")
        h = C.c_void_p()
        st = self.L.fpx_create(C.byref(cfg), C.byref(h))
        if st:
            raise FpxError(st, "fpx_create")
        self._h = h
        self.S, self.R = cfg.num_slots, cfg.num_replicas
        self.ngroups = cfg.num_groups * cfg.num_leader_groups

    def close(self):
        if getattr(self, "_h", None):
            self.L.fpx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- lifecycle -------------------------------------------------------------------------
    def reset(self):
        st = self.L.fpx_reset(self._h)
        if st:
            raise FpxError(st, "fpx_reset")

    def set_stream(self, hip_stream):
        """hip_stream: integer hipStream_t, e.g. torch.cuda.current_stream().cuda_stream (0 = the
        device's default stream); None selects the context's private stream (FPX_STREAM_OWN)"""
        st = self.L.fpx_set_stream(self._h, C.c_void_p(-1 if hip_stream is None else int(hip_stream)))
        if st:
            raise FpxError(st, "fpx_set_stream")

    def sync(self):
        """waits for the stream; returns the sticky device status of the _dev calls"""
        return self.L.fpx_sync(self._h)

    def error_detail(self):
        i, s, r = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.fpx_error_detail(self._h, C.byref(i), C.byref(s), C.byref(r))
        return i.value, s.value, r.value

    def profile_enable(self, on=True):
        st = self.L.fpx_profile_enable(self._h, int(on))
        if st:
            raise FpxError(st, "fpx_profile_enable")

    def profile_read(self):
        """(launches, total_ms) of the dominant kernel since the last read (HIP events on the stream)"""
        n, ms = C.c_int32(), C.c_double()
        st = self.L.fpx_profile_read(self._h, C.byref(n), C.byref(ms))
        if st:
            raise FpxError(st, "fpx_profile_read")
        return n.value, ms.value

    def profile_read_launches(self, cap=4096):
        """durations in ms of the timed launches since the last read, in launch order"""
        out = (C.c_float * cap)()
        n = C.c_int32()
        st = self.L.fpx_profile_read_launches(self._h, cap, out, C.byref(n))
        if st:
            raise FpxError(st, "fpx_profile_read_launches")
        return [float(out[i]) for i in range(min(cap, n.value))]

    @property
    def device_bytes(self):
        return self.L.fpx_device_bytes(self._h)

    def acceptor_max_voted_in(self, group, replica, first_slot=0, count=None):
        """Acceptor.maxVotedSlot over the slots [first_slot, first_slot + count) of the acceptor's group (-1: no vote there)"""
        out = C.c_int32(-1)
        st = self.L.fpx_acceptor_max_voted_in(self._h, group, replica, first_slot, self.cfg.num_slots - first_slot if count is None else count,
                                              C.byref(out))
        if st:
            raise FpxError(st, "fpx_acceptor_max_voted_in")
        return out.value

    def placement_stats(self):
        """how fpx_create placed the cell arrays: {"chunks": bool, "windows": n, "probe_ms": (min, median, max)}"""
        out = (C.c_float * 5)()
        st = self.L.fpx_placement_stats(self._h, out)
        if st:
            raise FpxError(st, "fpx_placement_stats")
        pr, un, ms = C.c_int32(0), C.c_int32(0), C.c_float(0)
        st = self.L.fpx_placement_search(self._h, C.byref(pr), C.byref(un), C.byref(ms))
        if st:
            raise FpxError(st, "fpx_placement_search")
        return {"chunks": bool(out[0]), "windows": int(out[1]), "probe_ms": (float(out[2]), float(out[3]), float(out[4])),
                "search": {"probes": pr.value, "unprobed_decisions": un.value, "ms": float(ms.value)}}

    def deferred_folds(self):
        """diagnostic: fused steps whose launch carried the fold of the step before (include/fpx.h)"""
        return int(self.L.fpx_deferred_folds(self._h))

    def band_merged_steps(self):
        """diagnostic: the mencius_band_fused_dev steps that ran in the two-launch form"""
        return int(self.L.fpx_band_merged_steps(self._h))

    # ---- host-pointer entry points (numpy) ---------------------------------------------------
    def acceptor_phase2a(self, slot, round_, value, target_mask=None):
        slot, round_, value, target_mask = _i32(slot), _i32(round_), _i32(value), _u64(target_mask)
        n = len(slot)
        vb = np.zeros((n, 4), np.uint64)
        nb = np.zeros((n, 4), np.uint64)
        nr = np.zeros(n, np.int32)
        st = self.L.fpx_acceptor_phase2a(self._h, n, _hp(slot), _hp(round_), _hp(value),
                                         _hp(target_mask), _hp(vb), _hp(nb), _hp(nr))
        return st, vb, nb, nr

    def acceptor_phase1a(self, group, round_, watermark=0, target_mask=None):
        target_mask = _u64(target_mask)
        pb = np.zeros(4, np.uint64)
        nb = np.zeros(4, np.uint64)
        st = self.L.fpx_acceptor_phase1a(self._h, group, round_, watermark, _hp(target_mask),
                                         _hp(pb), _hp(nb))
        return st, pb, nb

    def acceptor_phase1a_dev(self, group, round_, watermark=0, target_mask=None, promised_bits=None,
                             nack_bits=None):
        """asynchronous Phase1a: torch CUDA tensors of 4 int64 words each (or None)"""
        st = self.L.fpx_acceptor_phase1a_dev(self._h, group, round_, watermark, _dp(target_mask),
                                             _dp(promised_bits), _dp(nack_bits))
        if st:
            raise FpxError(st, "fpx_acceptor_phase1a_dev")

    def flush_promises(self):
        st = self.L.fpx_acceptor_flush_promises(self._h)
        if st:
            raise FpxError(st, "fpx_acceptor_flush_promises")

    def proxy_open(self, slot, round_, value):
        slot, round_, value = _i32(slot), _i32(round_), _i32(value)
        n = len(slot)
        new = np.zeros(n, np.uint8)
        st = self.L.fpx_proxy_open(self._h, n, _hp(slot), _hp(round_), _hp(value), _hp(new))
        return st, new

    def proxy_phase2b(self, slot, round_, vote_bits):
        slot, round_, vote_bits = _i32(slot), _i32(round_), _u64(vote_bits)
        n = len(slot)
        ch = np.zeros(n, np.uint8)
        cr = np.zeros(n, np.int32)
        cv = np.zeros(n, np.int32)
        st = self.L.fpx_proxy_phase2b(self._h, n, _hp(slot), _hp(round_), _hp(vote_bits), _hp(ch),
                                      _hp(cr), _hp(cv))
        return st, ch, cr, cv

    def phase2_fused(self, slot, round_, value, target_mask=None):
        slot, round_, value, target_mask = _i32(slot), _i32(round_), _i32(value), _u64(target_mask)
        n = len(slot)
        ch = np.zeros(n, np.uint8)
        cr = np.zeros(n, np.int32)
        cv = np.zeros(n, np.int32)
        nr = np.zeros(n, np.int32)
        st = self.L.fpx_phase2_fused(self._h, n, _hp(slot), _hp(round_), _hp(value),
                                     _hp(target_mask), _hp(ch), _hp(cr), _hp(cv), _hp(nr))
        return st, ch, cr, cv, nr

    # ---- device-pointer entry points (torch CUDA tensors, async) ------------------------------
    def acceptor_phase2a_dev(self, slot, round_, value, target_mask=None, vote_bits=None,
                             nack_bits=None, nack_round=None):
        st = self.L.fpx_acceptor_phase2a_dev(self._h, slot.numel(), _dp(slot), _dp(round_),
                                             _dp(value), _dp(target_mask), _dp(vote_bits),
                                             _dp(nack_bits), _dp(nack_round))
        if st:
            raise FpxError(st, "fpx_acceptor_phase2a_dev")

    def proxy_open_dev(self, slot, round_, value, is_new=None):
        st = self.L.fpx_proxy_open_dev(self._h, slot.numel(), _dp(slot), _dp(round_), _dp(value),
                                       _dp(is_new))
        if st:
            raise FpxError(st, "fpx_proxy_open_dev")

    def proxy_phase2b_dev(self, slot, round_, vote_bits, newly_chosen=None, chosen_round=None,
                          chosen_value=None):
        st = self.L.fpx_proxy_phase2b_dev(self._h, slot.numel(), _dp(slot), _dp(round_),
                                          _dp(vote_bits), _dp(newly_chosen), _dp(chosen_round),
                                          _dp(chosen_value))
        if st:
            raise FpxError(st, "fpx_proxy_phase2b_dev")

    def phase2_fused_dev(self, slot, round_, value, target_mask=None, chosen=None, chosen_round=None,
                         chosen_value=None, nack_round=None):
        st = self.L.fpx_phase2_fused_dev(self._h, slot.numel(), _dp(slot), _dp(round_), _dp(value),
                                         _dp(target_mask), _dp(chosen), _dp(chosen_round),
                                         _dp(chosen_value), _dp(nack_round))
        if st:
            raise FpxError(st, "fpx_phase2_fused_dev")

    # ---- the wire adapter on the device (include/fpx_wire.h) -------------------------------------------
    def wire_decode_dev(self, which, buf, offsets, value_id_base=0, buf_len=None):
        """which = "proxy_leader_inbound" | "acceptor_inbound"; buf: uint8 CUDA tensor holding the tick's messages back
        to back, offsets: int64 CUDA tensor [n + 1].  Returns a dict of CUDA tensors (the SoA batch), enqueued on the
        context's stream; errors surface at sync() like every _dev call."""
        import torch
        n = offsets.numel() - 1
        names = ["kind", "slot", "round", "is_noop", "value_off", "value_len"] + \
            (["group_index", "acceptor_index"] if which == "proxy_leader_inbound" else ["chosen_watermark"]) + ["value_id"]
        out = {k: torch.empty(max(n, 1), dtype=torch.int64 if k == "value_off" else torch.int32, device=buf.device)[:n]
               for k in names}
        from . import wire
        fn = getattr(wire._L(), "fpx_wire_decode_%s_dev" % which)
        st = fn(self._h, _dp(buf), buf.numel() if buf_len is None else buf_len, _dp(offsets), n,
                *[out[k].data_ptr() for k in names[:-1]], value_id_base, out["value_id"].data_ptr())
        if st:
            raise FpxError(st, "fpx_wire_decode_%s_dev" % which)
        return out

    # ---- multi-GPU: RCCL communicator behind the C ABI (fpx_comm_*) ----------------------------------
    def comm_create(self, unique_id, rank, world):
        """collective over the `world` contexts (one per GPU): unique_id = comm_unique_id() of one rank"""
        buf = (C.c_uint8 * _lib.FPX_COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        st = self.L.fpx_comm_create(self._h, buf, rank, world)
        if st:
            raise FpxError(st, "fpx_comm_create (rccl %d)" % self.L.fpx_last_rccl_error(self._h))

    def comm_destroy(self):
        st = self.L.fpx_comm_destroy(self._h)
        if st:
            raise FpxError(st, "fpx_comm_destroy")

    def comm_info(self):
        r, w = C.c_int32(), C.c_int32()
        self.L.fpx_comm_info(self._h, C.byref(r), C.byref(w))
        return r.value, w.value

    def phase2_replica_sharded_dev(self, slot, round_, value, target_mask=None, chosen=None,
                                   chosen_round=None, chosen_value=None, nack_round=None):
        """replica-axis sharded fused step: K1 on my acceptors -> ncclReduceScatter(sum) of the partial vote
        bitmaps -> open + K2 on my slice of the batch (outputs have n / world entries)"""
        st = self.L.fpx_phase2_replica_sharded_dev(self._h, slot.numel(), _dp(slot), _dp(round_), _dp(value),
                                                   _dp(target_mask), _dp(chosen), _dp(chosen_round),
                                                   _dp(chosen_value), _dp(nack_round))
        if st:
            raise FpxError(st, "fpx_phase2_replica_sharded_dev (rccl %d)" % self.L.fpx_last_rccl_error(self._h))

    def comm_allgather_chosen_dev(self, chosen, chosen_round, chosen_value, all_chosen, all_round, all_value):
        n = (chosen if chosen is not None else chosen_value).numel()
        st = self.L.fpx_comm_allgather_chosen_dev(self._h, n, _dp(chosen), _dp(chosen_round), _dp(chosen_value),
                                                  _dp(all_chosen), _dp(all_round), _dp(all_value))
        if st:
            raise FpxError(st, "fpx_comm_allgather_chosen_dev")

    def profile_read_collective(self):
        n, ms = C.c_int32(), C.c_double()
        st = self.L.fpx_profile_read_collective(self._h, C.byref(n), C.byref(ms))
        if st:
            raise FpxError(st, "fpx_profile_read_collective")
        return n.value, ms.value

    def proxy_forget(self, first_slot, count):
        """GC of the proxy leader's tallies of a slot range (async on the context's stream)"""
        st = self.L.fpx_proxy_forget(self._h, first_slot, count)
        if st:
            raise FpxError(st, "fpx_proxy_forget")

    def recycle_slots(self, first_slot, count):
        """the rows of a slot range become fresh: votes dropped, tallies forgotten, promises kept (async)"""
        st = self.L.fpx_recycle_slots(self._h, first_slot, count)
        if st:
            raise FpxError(st, "fpx_recycle_slots")

    # ---- K4: Mencius noop ranges ----------------------------------------------------------------
    def acceptor_phase2a_noop_range(self, slot_start, slot_end, round_, target_masks=None):
        A = self.cfg.num_groups
        target_masks = _u64(target_masks)
        vb = np.zeros((A, 4), np.uint64)
        nb = np.zeros((A, 4), np.uint64)
        nr = C.c_int32(-1)
        st = self.L.fpx_acceptor_phase2a_noop_range(self._h, slot_start, slot_end, round_,
                                                    _hp(target_masks), _hp(vb), _hp(nb), C.byref(nr))
        return st, vb, nb, nr.value

    def proxy_open_noop_range(self, slot_start, slot_end, round_):
        new = C.c_uint8(0)
        st = self.L.fpx_proxy_open_noop_range(self._h, slot_start, slot_end, round_, C.byref(new))
        return st, new.value

    def proxy_phase2b_noop_range(self, slot_start, slot_end, round_, vote_bits):
        vote_bits = _u64(vote_bits)
        ch = C.c_uint8(0)
        st = self.L.fpx_proxy_phase2b_noop_range(self._h, slot_start, slot_end, round_, _hp(vote_bits),
                                                 C.byref(ch))
        return st, ch.value

    # batched forms (n ranges per call; bitmaps n x num_groups x 4)
    def _ranges(self, slot_start, slot_end, round_):
        return _i32(np.atleast_1d(slot_start)), _i32(np.atleast_1d(slot_end)), _i32(np.atleast_1d(round_))

    def acceptor_phase2a_noop_ranges(self, slot_start, slot_end, round_, target_masks=None):
        s, e, r = self._ranges(slot_start, slot_end, round_)
        n, A = len(s), self.cfg.num_groups
        target_masks = _u64(target_masks)
        vb = np.zeros((n, A, 4), np.uint64)
        nb = np.zeros((n, A, 4), np.uint64)
        nr = np.full(n, -1, np.int32)
        st = self.L.fpx_acceptor_phase2a_noop_ranges(self._h, n, _hp(s), _hp(e), _hp(r), _hp(target_masks),
                                                     _hp(vb), _hp(nb), _hp(nr))
        return st, vb, nb, nr

    def proxy_open_noop_ranges(self, slot_start, slot_end, round_):
        s, e, r = self._ranges(slot_start, slot_end, round_)
        new = np.zeros(len(s), np.uint8)
        st = self.L.fpx_proxy_open_noop_ranges(self._h, len(s), _hp(s), _hp(e), _hp(r), _hp(new))
        return st, new

    def proxy_phase2b_noop_ranges(self, slot_start, slot_end, round_, vote_bits):
        s, e, r = self._ranges(slot_start, slot_end, round_)
        vote_bits = _u64(vote_bits)
        ch = np.zeros(len(s), np.uint8)
        st = self.L.fpx_proxy_phase2b_noop_ranges(self._h, len(s), _hp(s), _hp(e), _hp(r), _hp(vote_bits), _hp(ch))
        return st, ch

    def noop_ranges_fused(self, slot_start, slot_end, round_, target_masks=None):
        """open + acceptors + tally for n ranges: (status, vote_bits, nack_bits, nack_round, is_new, chosen)"""
        s, e, r = self._ranges(slot_start, slot_end, round_)
        n, A = len(s), self.cfg.num_groups
        target_masks = _u64(target_masks)
        vb = np.zeros((n, A, 4), np.uint64)
        nb = np.zeros((n, A, 4), np.uint64)
        nr = np.full(n, -1, np.int32)
        new = np.zeros(n, np.uint8)
        ch = np.zeros(n, np.uint8)
        st = self.L.fpx_noop_ranges_fused(self._h, n, _hp(s), _hp(e), _hp(r), _hp(target_masks), _hp(vb), _hp(nb),
                                          _hp(nr), _hp(new), _hp(ch))
        return st, vb, nb, nr, new, ch

    def noop_ranges_fused_dev(self, slot_start, slot_end, round_, target_masks=None, vote_bits=None,
                              nack_bits=None, nack_round=None, is_new=None, chosen=None):
        st = self.L.fpx_noop_ranges_fused_dev(self._h, slot_start.numel(), _dp(slot_start), _dp(slot_end),
                                              _dp(round_), _dp(target_masks), _dp(vote_bits), _dp(nack_bits),
                                              _dp(nack_round), _dp(is_new), _dp(chosen))
        if st:
            raise FpxError(st, "fpx_noop_ranges_fused_dev")

    def mencius_band_fused_dev(self, slot, round_, value, target_mask, chosen, chosen_round, chosen_value, nack_round,
                               slot_start, slot_end, range_round, range_target_masks=None, range_vote_bits=None,
                               range_nack_bits=None, range_nack_round=None, range_is_new=None, range_chosen=None,
                               independent=False):
        """one Mencius proxy-leader step: phase2_fused_dev on the commands + noop_ranges_fused_dev on the ranges; with
        independent (no leader group has both) and FPX_F_TRUSTED the step is two launches (or the halves side by side)"""
        st = self.L.fpx_mencius_band_fused_dev(
            self._h, slot.numel(), _dp(slot), _dp(round_), _dp(value), _dp(target_mask), _dp(chosen), _dp(chosen_round),
            _dp(chosen_value), _dp(nack_round), slot_start.numel(), _dp(slot_start), _dp(slot_end), _dp(range_round),
            _dp(range_target_masks), _dp(range_vote_bits), _dp(range_nack_bits), _dp(range_nack_round), _dp(range_is_new),
            _dp(range_chosen), 1 if independent else 0)
        if st:
            raise FpxError(st, "fpx_mencius_band_fused_dev")

    def read_range_tally(self, slot_start, slot_end, round_):
        state = C.c_int32()
        bits = np.zeros((self.cfg.num_groups, 4), np.uint64)
        st = self.L.fpx_read_range_tally(self._h, slot_start, slot_end, round_, C.byref(state), _hp(bits))
        if st:
            raise FpxError(st, "fpx_read_range_tally")
        return state.value, bits

    # ---- f1: replica log / f2: Phase-1 recovery scan ----------------------------------------------
    def replica_chosen(self, slot, value, mask=None):
        slot, value = _i32(slot), _i32(value)
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        wm, nc = C.c_int32(), C.c_int32()
        st = self.L.fpx_replica_chosen(self._h, len(slot), _hp(slot), _hp(value), _hp(mask),
                                       C.byref(wm), C.byref(nc))
        return st, wm.value, nc.value

    def replica_chosen_noop_range(self, slot_start, slot_end):
        wm, nc = C.c_int32(), C.c_int32()
        st = self.L.fpx_replica_chosen_noop_range(self._h, slot_start, slot_end, C.byref(wm), C.byref(nc))
        return st, wm.value, nc.value

    def replica_chosen_dev(self, slot, value, mask=None):
        st = self.L.fpx_replica_chosen_dev(self._h, slot.numel(), _dp(slot), _dp(value), _dp(mask))
        if st:
            raise FpxError(st, "fpx_replica_chosen_dev")

    def replica_state(self):
        wm, nc = C.c_int32(), C.c_int32()
        st = self.L.fpx_replica_state(self._h, C.byref(wm), C.byref(nc))
        if st:
            raise FpxError(st, "fpx_replica_state")
        return wm.value, nc.value

    def replica_read_log(self, first, count):
        vals = np.zeros(count, np.int32)
        pres = np.zeros(count, np.uint8)
        st = self.L.fpx_replica_read_log(self._h, first, count, _hp(vals), _hp(pres))
        if st:
            raise FpxError(st, "fpx_replica_read_log")
        return vals, pres

    def leader_phase1b_scan(self, watermark, quorum_masks, cap):
        q = np.ascontiguousarray(quorum_masks, dtype=np.uint64).reshape(self.ngroups, 4)
        mx = C.c_int32()
        sr = np.full(cap, -7, np.int32)
        sv = np.full(cap, -7, np.int32)
        st = self.L.fpx_leader_phase1b_scan(self._h, watermark, _hp(q), cap, C.byref(mx), _hp(sr),
                                            _hp(sv))
        k = max(0, min(cap, mx.value - watermark + 1))
        return st, mx.value, sr[:k], sv[:k]

    def acceptor_phase1b_info(self, group, replica, watermark=0):
        """Phase1b.info of one acceptor: (slot, vote_round, vote_value) of its votes in slots >= watermark, ascending"""
        k = C.c_int32()
        st = self.L.fpx_acceptor_phase1b_info(self._h, group, replica, watermark, 0, C.byref(k), None, None, None)
        if st:
            raise FpxError(st, "fpx_acceptor_phase1b_info")
        n = k.value
        sl, vr, vv = (np.zeros(n, np.int32) for _ in range(3))
        if n:
            st = self.L.fpx_acceptor_phase1b_info(self._h, group, replica, watermark, n, C.byref(k), _hp(sl), _hp(vr), _hp(vv))
            if st:
                raise FpxError(st, "fpx_acceptor_phase1b_info")
        return sl, vr, vv

    # ---- readback ------------------------------------------------------------------------------
    def read_acceptor(self, group, replica):
        p, m = C.c_int32(), C.c_int32()
        vr = np.zeros(self.S, np.int32)
        vv = np.zeros(self.S, np.int32)
        bl = np.zeros(self.S, np.int32)
        st = self.L.fpx_read_acceptor(self._h, group, replica, C.byref(p), C.byref(m), _hp(vr),
                                      _hp(vv), _hp(bl))
        if st:
            raise FpxError(st, "fpx_read_acceptor")
        return p.value, m.value, vr, vv, bl

    def read_state(self):
        vr = np.zeros((self.S, self.R), np.int32)
        vv = np.zeros((self.S, self.R), np.int32)
        bl = np.zeros((self.S, self.R), np.int32)
        st = self.L.fpx_read_state(self._h, _hp(vr), _hp(vv), _hp(bl))
        if st:
            raise FpxError(st, "fpx_read_state")
        return vr, vv, bl

    def read_scalars(self):
        pr = np.zeros((self.ngroups, self.R), np.int32)
        mv = np.zeros((self.ngroups, self.R), np.int32)
        st = self.L.fpx_read_scalars(self._h, _hp(pr), _hp(mv))
        if st:
            raise FpxError(st, "fpx_read_scalars")
        return pr, mv

    def state_digest(self):
        """8 uint64 digests of the whole state (fpx_state_digest): equal states have equal digests; the converse holds up
        to a collision of a 64-bit additive hash of (position, value) terms -- improbable, not impossible"""
        out = np.zeros(8, np.uint64)
        st = self.L.fpx_state_digest(self._h, _hp(out))
        if st:
            raise FpxError(st, "fpx_state_digest")
        return out

    def read_tally(self, slot):
        n = C.c_int32()
        rounds = np.zeros(8, np.int32)
        states = np.zeros(8, np.int32)
        values = np.zeros(8, np.int32)
        bits = np.zeros((8, 4), np.uint64)
        st = self.L.fpx_read_tally(self._h, slot, C.byref(n), _hp(rounds), _hp(states), _hp(values),
                                   _hp(bits))
        if st:
            raise FpxError(st, "fpx_read_tally")
        return [(int(rounds[i]), int(states[i]), int(values[i]), tuple(int(x) for x in bits[i]))
                for i in range(n.value)]


def comm_unique_id():
    """ncclGetUniqueId through the C ABI: 128 opaque bytes for fpx_comm_create on every rank"""
    buf = (C.c_uint8 * _lib.FPX_COMM_ID_BYTES)()
    st = _lib.lib().fpx_comm_unique_id(buf)
    if st:
        raise FpxError(st, "fpx_comm_unique_id")
    return bytes(buf)


# ---- a5 / a7 free functions ----------------------------------------------------------------------
class PinnedArray:
    """A numpy array over page-locked host memory (fpx_host_alloc): batches built in it cross PCIe by
    DMA.  Keep the object alive while `array` is in use; `free()` (or garbage collection) releases it."""

    def __init__(self, shape, dtype):
        self._L = _lib.lib()
        dtype = np.dtype(dtype)
        nbytes = max(1, int(np.prod(shape)) * dtype.itemsize)
        p = C.c_void_p()
        st = self._L.fpx_host_alloc(nbytes, C.byref(p))
        if st:
            raise FpxError(st, "fpx_host_alloc")
        self._p = p
        buf = (C.c_char * nbytes).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._p is not None:
            self.array = None
            self._L.fpx_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def quorum_eval(cfg, nodes, strict=True, read=False):
    """isWriteQuorum / isReadQuorum (strict) or the isSuperSetOf* variants for n node sets
    (n x 4 uint64), evaluated by the device predicate the tally kernels use."""
    L = _lib.lib()
    nodes = np.ascontiguousarray(nodes, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(len(nodes), np.uint8)
    fn = L.fpx_read_quorum_eval if read else L.fpx_quorum_eval
    st = fn(C.byref(cfg), len(nodes), _hp(nodes), int(strict), _hp(out))
    if st == _lib.FPX_EINVAL:
        raise ValueError("IllegalArgumentException (require failed)")
    if st:
        raise FpxError(st, "fpx_quorum_eval")
    return out.astype(bool)


def round_leader(num_leaders, round_):
    return _lib.lib().fpx_round_leader(num_leaders, round_)


def next_classic_round(num_leaders, leader_index, round_):
    return _lib.lib().fpx_next_classic_round(num_leaders, leader_index, round_)
