#!/usr/bin/env python3
"""Generates the committed golden fixtures tests/golden/*.npz.

The reference (Scala/JVM) cannot run in this environment, so these vectors are produced by the CPU
oracle (oracle/fpx_oracle.c) -- the line-by-line restatement of the reference handlers that
tests/test_oracle_golden.py pins on the reference's own known-answer tests.  They freeze the oracle's
answers on the seeded streams of SURVEY.md section 8(d), so that (a) the oracle cannot drift silently
and (b) the HIP path is checked against committed vectors, not only against a live oracle.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as O  # noqa: E402
from tests import workloads as W  # noqa: E402

CASES = {
    # name: (config kwargs, script kwargs)
    "adv_r256_s1": (dict(num_slots=512, num_replicas=256, f=127, tally_ways=8), dict(R=256, q=128, seed=1, epochs=16, fused=True)),
    "adv_r256_s2_perslot": (dict(num_slots=512, num_replicas=256, f=127, ballot_mode=1, tally_ways=8), dict(R=256, q=128, seed=2, epochs=16, fused=True)),
    "adv_r256_s3_unfused": (dict(num_slots=512, num_replicas=256, f=127, tally_ways=8), dict(R=256, q=128, seed=3, epochs=16, fused=False)),
    "adv_r3_f1": (dict(num_slots=2048, num_replicas=3, f=1, tally_ways=8), dict(R=3, q=2, seed=4, epochs=32, fused=True)),
    "adv_grid2x2_16groups": (dict(num_slots=2048, num_replicas=4, num_groups=16, quorum_kind=2, grid_rows=2, grid_cols=2, tally_ways=8),
                             dict(R=4, q=2, seed=5, epochs=16, fused=True, ngroups=16)),
    "adv_mencius_8x2": (dict(num_slots=2048, num_replicas=3, num_groups=2, num_leader_groups=8, f=1, tally_ways=8),
                        dict(R=3, q=2, seed=6, epochs=16, fused=False, ngroups=16)),
    "adv_r100_majority": (dict(num_slots=512, num_replicas=100, quorum_kind=1, ballot_mode=1, tally_ways=8), dict(R=100, q=51, seed=7, epochs=16, fused=True)),
}


def flatten(outputs):
    """every array / scalar of every op output, concatenated as int64"""
    parts = []
    for out in outputs:
        for x in out[1:]:
            a = np.asarray(x)
            if a.dtype == np.uint64:
                a = a.view(np.int64)
            parts.append(a.astype(np.int64).ravel())
    return np.concatenate(parts)


def run_case(be, cfg_kw, script_kw):
    S = cfg_kw["num_slots"]
    script = W.adversarial_script(S, script_kw["R"], script_kw["q"], script_kw["seed"], epochs=script_kw["epochs"],
                                  fused=script_kw["fused"], ngroups=script_kw.get("ngroups", 1))
    outs = W.run_script(be, script)
    snap = W.snapshot(be)
    return {"outputs": flatten(outs), **{k: v for k, v in snap.items()}}


def main():
    O.build()
    for name, (cfg_kw, script_kw) in CASES.items():
        be = O.System(O.make_config(**cfg_kw))
        data = run_case(be, cfg_kw, script_kw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print("%-28s outputs %8d ints, state %s" % (name, len(data["outputs"]), data["vote_round"].shape))


if __name__ == "__main__":
    main()
