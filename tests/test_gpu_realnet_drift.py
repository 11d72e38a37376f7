"""-m gpu: the search with the REAL network, end to end, against the reference's torch-CPU run of the same seeds (gv14).

north_star asks for visit counts and chosen moves bit-exact under a fixed seed AND for policy / value within 1e-4: the two meet
only where the evaluations are bit-identical, which is how tree parity is defined and tested (gv5 / gv6: replayed (p, v)). This
test states what is left when the evaluations are the native forward's (agents.py:170-221 consumes them as they are) -- measured
by tools/realnet_visit_drift.py, numbers in DESIGN.md section 2."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_native_network_search_against_the_reference_torch_cpu_run(capsys):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import realnet_visit_drift as D
    rep = D.measure()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "realnet_visit_drift.json"), "w") as f:
        json.dump(rep, f, indent=1)
    with capsys.disabled():
        for r in rep:
            print("\n  %s, seed %d, net mode %d: first ply parted %r; " % (r["network"], r["seed"], r["net_mode"], r["first_ply_parted"]) +
                  " ".join("%d:%s%s%s/%.2f" % (p["ply"], "V" if p["visits_equal"] else "v", "A" if p["action_equal"] else "a",
                                               "P" if p["mt_pos_equal"] else "p", p["visit_mass_on_the_same_moves"]) for p in r["plies"]), end="")
        print()
    assert len(rep) == 12
    # Measured (round 6, MI355X): all 60 searches on 9x9 -- 24 000 simulations, random-init and trained network, split-fp16 and fp32-MFMA
    # forward -- and the 15x15 / 10-block case reproduce the reference's visit vectors, moves and stream positions EXACTLY: evaluations within 1e-5 of torch's
    # were never close enough to a PUCT tie to order two children differently. That is an observation about these searches, not a
    # guarantee (a different summation order in any kernel may flip one); it is asserted so that such a change shows up here.
    for r in rep:
        assert r["first_ply_parted"] is None and len(r["plies"]) == r["recorded_plies"], (r["network"], r["seed"], r["net_mode"], r["first_ply_parted"])
        assert all(p["visits_equal"] and p["action_equal"] and p["mt_pos_equal"] for p in r["plies"])
    for r in rep:
        p0 = r["plies"][0]
        # what always holds: every search ran its simulations on the reference's position, from the reference's stream
        assert p0["visits"] == p0["ref_visits"] == r["sims"]
        for prev, p in zip(r["plies"], r["plies"][1:]):
            if prev["visits_equal"]:                      # the same tree was inherited
                assert p["visits"] == p["ref_visits"]
