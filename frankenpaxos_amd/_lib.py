"""ctypes binding of libfpx.so (the C ABI of include/fpx.h).

The library is built in-tree (frankenpaxos_amd/csrc/libfpx.so, see __graft_entry__.build()).  If it
is missing, or the GPU is missing, everything here fails loudly: there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("FPX_LIB") or os.path.join(CSRC, "libfpx.so")  # FPX_LIB: tuning builds

FPX_OK = 0
FPX_EINVAL = 1
FPX_EFATAL_UNKNOWN_SLOTROUND = 2
FPX_EHIP = 3
FPX_ENODEVICE = 4
FPX_ECAPACITY = 5
FPX_EORDER = 6
FPX_ENOMEM = 7
FPX_ERCCL = 8
FPX_EFATAL_PROTOCOL = 9
FPX_COMM_ID_BYTES = 128

FPX_Q_THRESHOLD = 0
FPX_Q_SIMPLE_MAJORITY = 1
FPX_Q_GRID = 2
FPX_Q_UNANIMOUS = 3

FPX_BALLOT_ACCEPTOR = 0
FPX_BALLOT_PER_SLOT = 1

FPX_F_TRUSTED = 1
FPX_F_SCATTERED_TARGETS = 2
FPX_F_SLOT_MAJOR_ROWS = 4

FPX_NOOP = -1


class FpxConfig(C.Structure):
    """fpx_config (include/fpx.h)."""

    _fields_ = [
        ("num_slots", C.c_int32),
        ("num_replicas", C.c_int32),
        ("num_groups", C.c_int32),
        ("num_leader_groups", C.c_int32),
        ("f", C.c_int32),
        ("quorum_kind", C.c_int32),
        ("grid_rows", C.c_int32),
        ("grid_cols", C.c_int32),
        ("num_leaders", C.c_int32),
        ("ballot_mode", C.c_int32),
        ("tally_ways", C.c_int32),
        ("replica_base", C.c_int32),
        ("replicas_total", C.c_int32),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
    ]


I32P = C.POINTER(C.c_int32)
U64P = C.POINTER(C.c_uint64)
U8P = C.POINTER(C.c_uint8)
CFGP = C.POINTER(FpxConfig)
VP = C.c_void_p

# every symbol include/fpx.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "fpx_version": (C.c_int32, []),
    "fpx_strerror": (C.c_char_p, [C.c_int32]),
    "fpx_config_check": (C.c_int32, [CFGP]),
    "fpx_create": (C.c_int32, [CFGP, C.POINTER(VP)]),
    "fpx_destroy": (C.c_int32, [VP]),
    "fpx_reset": (C.c_int32, [VP]),
    "fpx_set_stream": (C.c_int32, [VP, VP]),
    "fpx_sync": (C.c_int32, [VP]),
    "fpx_error_detail": (C.c_int32, [VP, I32P, I32P, I32P]),
    "fpx_last_hip_error": (C.c_int32, [VP]),
    "fpx_device_bytes": (C.c_int64, [VP]),
    "fpx_placement_stats": (C.c_int32, [VP, C.POINTER(C.c_float)]),
    "fpx_placement_search": (C.c_int32, [VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "fpx_band_merged_steps": (C.c_int64, [VP]),
    "fpx_deferred_folds": (C.c_int64, [VP]),
    "fpx_acceptor_max_voted_in": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, I32P]),
    "fpx_profile_read_launches": (C.c_int32, [VP, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "fpx_get_config": (C.c_int32, [VP, CFGP]),
    "fpx_host_alloc": (C.c_int32, [C.c_int64, C.POINTER(C.c_void_p)]),
    "fpx_host_free": (C.c_int32, [VP]),
    "fpx_profile_enable": (C.c_int32, [VP, C.c_int32]),
    "fpx_profile_read": (C.c_int32, [VP, I32P, C.POINTER(C.c_double)]),
    "fpx_round_leader": (C.c_int32, [C.c_int32, C.c_int32]),
    "fpx_next_classic_round": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "fpx_quorum_eval": (C.c_int32, [CFGP, C.c_int32, VP, C.c_int32, VP]),
    "fpx_is_write_quorum": (C.c_int32, [CFGP, VP, C.c_int32, VP]),
    "fpx_read_quorum_eval": (C.c_int32, [CFGP, C.c_int32, VP, C.c_int32, VP]),
    "fpx_acceptor_phase2a": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP, VP]),
    "fpx_acceptor_phase2a_dev": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP, VP]),
    "fpx_acceptor_phase1a": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, VP, VP, VP]),
    "fpx_acceptor_phase1a_dev": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, VP, VP, VP]),
    "fpx_acceptor_flush_promises": (C.c_int32, [VP]),
    "fpx_proxy_open": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP]),
    "fpx_proxy_open_dev": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP]),
    "fpx_proxy_phase2b": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP]),
    "fpx_proxy_phase2b_dev": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP]),
    "fpx_phase2_fused": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP, VP, VP]),
    "fpx_phase2_fused_submit": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP, VP, VP, I32P]),
    "fpx_phase2_fused_wait": (C.c_int32, [VP, C.c_int32]),
    "fpx_phase2_fused_dev": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP, VP, VP]),
    "fpx_acceptor_phase2a_noop_range": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, VP, VP, VP, I32P]),
    "fpx_proxy_open_noop_range": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, U8P]),
    "fpx_proxy_phase2b_noop_range": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, VP, U8P]),
    "fpx_acceptor_phase2a_noop_ranges": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP, VP]),
    "fpx_proxy_open_noop_ranges": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP]),
    "fpx_proxy_phase2b_noop_ranges": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP]),
    "fpx_noop_ranges_fused": (C.c_int32, [VP, C.c_int32] + [VP] * 9),
    "fpx_noop_ranges_fused_dev": (C.c_int32, [VP, C.c_int32] + [VP] * 9),
    "fpx_mencius_band_fused_dev": (C.c_int32, [VP, C.c_int32] + [VP] * 8 + [C.c_int32] + [VP] * 9 + [C.c_int32]),
    "fpx_read_range_tally": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, I32P, VP]),
    "fpx_recycle_slots": (C.c_int32, [VP, C.c_int32, C.c_int32]),
    "fpx_proxy_forget": (C.c_int32, [VP, C.c_int32, C.c_int32]),
    "fpx_epx_create": (C.c_int32, [VP, C.POINTER(VP)]),
    "fpx_epx_destroy": (C.c_int32, [VP]),
    "fpx_epx_info": (C.c_int32, [VP, I32P, I32P, I32P]),
    "fpx_epx_set_stream": (C.c_int32, [VP, VP]),
    "fpx_epx_sync": (C.c_int32, [VP]),
    "fpx_epx_preaccept": (C.c_int32, [VP, C.c_int32] + [VP] * 12),
    "fpx_epx_preaccept_dev": (C.c_int32, [VP, C.c_int32] + [VP] * 12),
    "fpx_epx_preaccept_packed_dev": (C.c_int32, [VP, C.c_int32] + [VP] * 9),
    "fpx_epx_packed_stride": (C.c_int32, [C.c_int32]),
    "fpx_epx_execute_dev": (C.c_int32, [VP, C.c_int32] + [VP] * 11),
    "fpx_epx_execute": (C.c_int32, [VP, C.c_int32] + [VP] * 12),
    "fpx_epx_handle_prepare_oks": (C.c_int32, [VP, C.c_int32] + [VP] * 8 + [C.c_int32] + [VP] * 3),
    "fpx_epx_prepare": (C.c_int32, [VP, C.c_int32] + [VP] * 12),
    "fpx_epx_accept": (C.c_int32, [VP, C.c_int32] + [VP] * 13),
    "fpx_epx_handle_commit": (C.c_int32, [VP, C.c_int32] + [VP] * 8),
    "fpx_epx_read_cmdlog": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, VP]),
    "fpx_epx_read_cmdlog_deps": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, VP, VP]),
    "fpx_epx_handle_preaccept": (C.c_int32, [VP, C.c_int32] + [VP] * 18),
    "fpx_epx_read_index": (C.c_int32, [VP, C.c_int32, C.c_int32, VP, VP]),
    "fpx_replica_chosen": (C.c_int32, [VP, C.c_int32, VP, VP, VP, I32P, I32P]),
    "fpx_replica_chosen_dev": (C.c_int32, [VP, C.c_int32, VP, VP, VP]),
    "fpx_replica_state": (C.c_int32, [VP, I32P, I32P]),
    "fpx_replica_chosen_noop_range": (C.c_int32, [VP, C.c_int32, C.c_int32, I32P, I32P]),
    "fpx_replica_read_log": (C.c_int32, [VP, C.c_int32, C.c_int32, VP, VP]),
    "fpx_leader_phase1b_scan": (C.c_int32, [VP, C.c_int32, VP, C.c_int32, I32P, VP, VP]),
    "fpx_acceptor_phase1b_info": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, I32P, VP, VP, VP]),
    "fpx_read_acceptor": (C.c_int32, [VP, C.c_int32, C.c_int32, I32P, I32P, VP, VP, VP]),
    "fpx_read_state": (C.c_int32, [VP, VP, VP, VP]),
    "fpx_read_scalars": (C.c_int32, [VP, VP, VP]),
    "fpx_read_tally": (C.c_int32, [VP, C.c_int32, I32P, VP, VP, VP, VP]),
    "fpx_state_digest": (C.c_int32, [VP, VP]),
    "fpx_comm_unique_id": (C.c_int32, [VP]),
    "fpx_comm_create": (C.c_int32, [VP, VP, C.c_int32, C.c_int32]),
    "fpx_comm_destroy": (C.c_int32, [VP]),
    "fpx_comm_info": (C.c_int32, [VP, I32P, I32P]),
    "fpx_last_rccl_error": (C.c_int32, [VP]),
    "fpx_phase2_replica_sharded_dev": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP, VP, VP]),
    "fpx_comm_allgather_chosen_dev": (C.c_int32, [VP, C.c_int32, VP, VP, VP, VP, VP, VP]),
    "fpx_profile_read_collective": (C.c_int32, [VP, I32P, C.POINTER(C.c_double)]),
}


class FpxError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = "libfpx status %d" % status
        try:
            msg += " (%s)" % lib().fpx_strerror(status).decode()
        except Exception:
            pass
        if what:
            msg += ": " + what
        super().__init__(msg)


def build(verbose=False):
    """Compile libfpx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "libfpx.so"], stdout=out)
    return SO_PATH


_lib = None


def lib():
    """Load libfpx.so; raises if the extension has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "libfpx.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
            "-- frankenpaxos_amd has no CPU fallback" % SO_PATH)
    # If torch is (going to be) in the process, make sure its bundled libamdhip64.so.7 is the HIP
    # runtime both sides share: import torch first, then libfpx resolves the SONAME to that copy.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
