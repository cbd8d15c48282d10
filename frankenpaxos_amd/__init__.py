"""frankenpaxos_amd: MI355X-native Phase-2 accept / quorum-tally engine.

The product is csrc/libfpx.so (hand-written HIP for gfx950 behind the C ABI of include/fpx.h);
this package is its thin Python binding.  There is no CPU implementation here: importing works
without a GPU, but creating a Context or evaluating a quorum without libfpx.so + a gfx950 device
raises.
"""
from ._lib import (FPX_BALLOT_ACCEPTOR, FPX_BALLOT_PER_SLOT, FPX_ECAPACITY, FPX_EFATAL_UNKNOWN_SLOTROUND,
                   FPX_EHIP, FPX_EINVAL, FPX_ENODEVICE, FPX_ENOMEM, FPX_EORDER, FPX_ERCCL, FPX_EFATAL_PROTOCOL, FPX_COMM_ID_BYTES, FPX_F_SCATTERED_TARGETS, FPX_F_SLOT_MAJOR_ROWS, FPX_F_TRUSTED, FPX_NOOP,
                   FPX_OK, FPX_Q_GRID, FPX_Q_SIMPLE_MAJORITY, FPX_Q_THRESHOLD, FPX_Q_UNANIMOUS, FpxConfig,
                   FpxError, build, lib)
from .context import Context, PinnedArray, comm_unique_id, make_config, next_classic_round, quorum_eval, round_leader

__all__ = [
    "Context", "make_config", "quorum_eval", "round_leader", "next_classic_round", "FpxConfig",
    "FpxError", "build", "lib",
]
