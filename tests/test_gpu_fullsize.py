"""GPU parity at the sizes BASELINE.json's configs state: the HIP path through the C ABI
against the CPU oracle, bit-exact on EVERY output of every op of the stream, plus whole-state digests
(fpx_state_digest vs the oracle's fpo_state_digest: vote rounds, vote values, ballots, acceptor scalars,
proxy-leader tallies -- equal digests <=> equal state) and sampled direct readbacks.

Run on the MI355X box: python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()
    return frankenpaxos_amd


def run_both(fa, oracle, script, sample_slots, **kw):
    gpu = fa.Context(fa.make_config(**kw))
    ref = oracle.System(oracle.make_config(**kw))
    out_g = W.run_script(gpu, script)
    out_r = W.run_script(ref, script)
    W.assert_same_outputs(out_g, out_r)
    dg, dr = gpu.state_digest(), ref.state_digest()
    np.testing.assert_array_equal(dg, dr, err_msg="state digests (vote_round, vote_value, ballot, promised, "
                                                  "max_voted, tallies, log)")
    pg, mg = gpu.read_scalars()
    pr, mr = ref.read_scalars()
    np.testing.assert_array_equal(pg, pr)
    np.testing.assert_array_equal(mg, mr)
    for s in sample_slots:
        assert gpu.read_tally(int(s)) == ref.read_tally(int(s)), "tally of slot %d" % s
    for g, r in ((0, 0), (gpu.ngroups - 1, gpu.R - 1)):
        a, b = gpu.read_acceptor(g, r), ref.read_acceptor(g, r)
        assert a[:2] == b[:2]
        for x, y in zip(a[2:], b[2:]):
            np.testing.assert_array_equal(x, y)
    chosen = sum(int(o[2].sum()) for o in out_g if o[0] == "fused") + \
        sum(int(o[8].sum()) for o in out_g if o[0] == "k1k2")
    gpu.close()
    return out_g, chosen


# ---------------------------------------------------------------------------------------------------
# the headline grid: 2^20 slots x 256 acceptors, adversarial stream of SURVEY.md 8(d), seeds 1-3
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_headline_adversarial_1m_slots_256_replicas(fa, oracle, seed, ballot_mode):
    S, R = 1 << 20, 256
    script = W.adversarial_script(S, R, 128, seed, epochs=64, fused=True, subsets=W.fast_subsets)
    out, chosen = run_both(fa, oracle, script, range(0, S, 65537), num_slots=S, num_replicas=R, f=127,
                           ballot_mode=ballot_mode, tally_ways=8)
    nacks = sum(int((o[5] >= 0).sum()) for o in out if o[0] == "fused")
    assert chosen >= S and nacks > 0          # every slot got chosen at least once; stale rounds were Nacked


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: compartmentalized MultiPaxos, 2x2 grid quorums, 1M slots x 16 acceptor groups
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ballot_mode,fused", [(0, True), (1, True), (0, False)])
def test_config3_1m_slots_16_groups_of_2x2_grids(fa, oracle, ballot_mode, fused):
    S = 1 << 20
    script = W.adversarial_script(S, 4, 2, 33 + ballot_mode, epochs=64, fused=fused, ngroups=16,
                                  subsets=W.fast_subsets)
    out, chosen = run_both(fa, oracle, script, range(0, S, 65521), num_slots=S, num_replicas=4, num_groups=16,
                           quorum_kind=2, grid_rows=2, grid_cols=2, ballot_mode=ballot_mode, tally_ways=8)
    # a 2x2 grid write quorum needs one acceptor of each row: some random target subsets miss a row
    assert 0 < chosen


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] at its stated size, unfused pipeline too (the fused one is in test_gpu_parity.py)
# ---------------------------------------------------------------------------------------------------
def test_config2_64k_slots_3_acceptors_adversarial(fa, oracle):
    S = 65536
    for fused in (True, False):
        script = W.adversarial_script(S, 3, 2, 17, epochs=64, fused=fused)
        run_both(fa, oracle, script, range(0, S, 4099), num_slots=S, num_replicas=3, f=1, tally_ways=8)
