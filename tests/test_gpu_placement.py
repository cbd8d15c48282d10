"""fpx_create's chunk placement of the cell slab (profiles/r05_placement.md; csrc/fpx_api.hip: place_chunks): a big context's
slab is a reserved address range backed by 1 GiB physical chunks whose partners are chosen by measurement.  The kernels
must see ONE contiguous slab: results, state digests and read-backs are those of a context whose slab is one allocation
(FPX_PLACEMENT_CHUNKS=0), the memory comes back when the context goes, and small contexts are left alone."""
import numpy as np
import pytest

from tests import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import frankenpaxos_amd

    frankenpaxos_amd.lib()
    return frankenpaxos_amd


@pytest.mark.parametrize("ballot_mode", [0, 1])
def test_chunked_slab_is_one_contiguous_slab_to_the_kernels(fa, oracle, monkeypatch, ballot_mode):
    import torch

    S, R = 1 << 20, 256                                   # 2 GiB (round per acceptor) / 3 GiB (ballot per cell) of cells
    kw = dict(num_slots=S, num_replicas=R, f=127, ballot_mode=ballot_mode, tally_ways=8)
    script = W.adversarial_script(S, R, 128, 21 + ballot_mode, epochs=16, fused=True, subsets=W.fast_subsets)
    free0 = torch.cuda.mem_get_info()[0]
    monkeypatch.delenv("FPX_PLACEMENT_CHUNKS", raising=False)
    a = fa.Context(fa.make_config(**kw))
    pa = a.placement_stats()
    assert pa["chunks"] and pa["windows"] == 1 and 0 < pa["probe_ms"][0] <= pa["probe_ms"][2] < 1.0, pa
    monkeypatch.setenv("FPX_PLACEMENT_CHUNKS", "0")
    b = fa.Context(fa.make_config(**kw))
    assert not b.placement_stats()["chunks"]
    out_a, out_b = W.run_script(a, script), W.run_script(b, script)
    W.assert_same_outputs(out_a, out_b)
    np.testing.assert_array_equal(a.state_digest(), b.state_digest())
    # rows on both sides of the chunk seams of every array, read back through the column gather
    for r in (0, 255):
        x, y = a.read_acceptor(0, r), b.read_acceptor(0, r)
        assert x[:2] == y[:2]
        for u, v in zip(x[2:], y[2:]):
            np.testing.assert_array_equal(u, v)
    # the oracle on a sample of the stream's first epoch (the full-size comparison is tests/test_gpu_fullsize.py's)
    ref = oracle.System(oracle.make_config(**kw))
    W.assert_same_outputs(W.run_script(ref, script[:3]), out_a[:3])
    a.reset()
    assert a.sync() == 0 and a.placement_stats()["chunks"]          # a reset keeps the slab
    a.close()
    b.close()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (256 << 20), "device memory of the two contexts did not come back"


def test_chunk_placement_survives_many_contexts_and_leaves_small_ones_alone(fa):
    import torch

    small = fa.Context(fa.make_config(num_slots=1 << 16, num_replicas=256, f=127, ballot_mode=1))
    assert not small.placement_stats()["chunks"]                      # 192 MiB of cells: one allocation
    narrow = fa.Context(fa.make_config(num_slots=1 << 24, num_replicas=3, f=1))
    assert not narrow.placement_stats()["chunks"]                     # 16-byte rows: the pairing is about 1 KiB rows
    small.close(), narrow.close()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(6):                                                # reserve / create / map / unmap / release, again and again
        c = fa.Context(fa.make_config(num_slots=(2 + i % 2) << 20, num_replicas=255, f=127, ballot_mode=i % 2, flags=fa.FPX_F_TRUSTED))
        st = c.placement_stats()
        assert st["chunks"] and st["windows"] >= 2
        slot, rnd, val = W.steady_stream(1 << 12)
        assert c.acceptor_phase1a(0, 0)[0] == 0
        res = c.phase2_fused(slot + (1 << 20), rnd, val)             # rows of the second gigabyte of every array
        assert res[0] == 0 and res[1].all() and (res[3] == val).all()
        c.close()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (256 << 20)


def test_placement_search_keeps_to_its_budget(fa, monkeypatch):
    """FPX_PLACEMENT_BUDGET_MS bounds the probe time of fpx_create -- with the budget spent every
    remaining decision takes its first candidate unprobed (budget 0: the slab is the chunks in allocation order), results
    are those of any other placement, and the default budget is reported kept on a quiet device."""
    kw = dict(num_slots=3 << 20, num_replicas=256, f=127, ballot_mode=1, flags=fa.FPX_F_TRUSTED)
    monkeypatch.delenv("FPX_PLACEMENT_CHUNKS", raising=False)
    monkeypatch.setenv("FPX_PLACEMENT_BUDGET_MS", "0")
    c = fa.Context(fa.make_config(**kw))
    st = c.placement_stats()
    assert st["chunks"] and st["windows"] == 3, st
    assert st["search"]["unprobed_decisions"] >= 3 and st["search"]["probes"] <= 2, st      # (one warm-up probe)
    slot, rnd, val = W.steady_stream(1 << 12)
    assert c.acceptor_phase1a(0, 0)[0] == 0
    for base in (0, 1 << 20, (3 << 20) - (1 << 12)):                   # rows of every gigabyte window
        res = c.phase2_fused(slot + base, rnd, val)
        assert res[0] == 0 and res[1].all() and (res[3] == val).all()
    c.close()
    monkeypatch.delenv("FPX_PLACEMENT_BUDGET_MS")
    c = fa.Context(fa.make_config(**kw))
    st = c.placement_stats()
    # the search of a 9-chunk slab: tens of probes of ~0.2 ms each; the wall clock stays near the default budget of 300 ms
    # (one probe may be in flight when it runs out)
    assert st["chunks"] and st["search"]["probes"] > 3 and st["search"]["ms"] < 600, st
    c.close()
