"""The reference's safety invariant under random delivery orders,
re-implemented for the oracle: see tests/paxos_sim.py.  CPU only; the GPU twin replays the same
schedules in tests/test_gpu_parity.py."""
import pytest

from tests import paxos_sim


@pytest.mark.parametrize("R,f", [(3, 1), (5, 2)])
def test_safety_under_random_schedules(oracle, R, f):
    total_chosen = 0
    for seed in range(40):
        be = oracle.System(oracle.make_config(num_slots=6, num_replicas=R, f=f, tally_ways=8))
        trace, chosen = paxos_sim.simulate(be, seed, R=R, f=f, S=6, steps=500)
        total_chosen += len(chosen)
        # acceptor-local invariant: a vote never exceeds the acceptor's round
        vr, _, _ = be.read_state()
        pr, _ = be.read_scalars()
        assert (vr <= pr[0][None, :]).all()
    assert total_chosen > 40  # the schedules do make progress


def test_a_protocol_without_promises_is_caught(oracle):
    """Sanity of the harness itself: if Phase1a promises are never recorded (PER_SLOT ballots with a
    watermark beyond the log, so no cell is promised), stale leaders overwrite newer votes and the
    safety check must fire within a few schedules."""
    import numpy as np

    violated = 0
    for seed in range(60):
        be = oracle.System(oracle.make_config(num_slots=4, num_replicas=3, f=1, tally_ways=8,
                                              ballot_mode=1))
        orig = be.acceptor_phase1a
        be.acceptor_phase1a = lambda g, rnd, wm=0, tgt=None, _o=orig: _o(g, rnd, 99, tgt)
        be.read_scalars = lambda: (np.full((1, 3), -1, np.int32), np.full((1, 3), -1, np.int32))
        try:
            paxos_sim.simulate(be, seed, R=3, f=1, S=4, steps=500)
        except AssertionError:
            violated += 1
    assert violated > 0
