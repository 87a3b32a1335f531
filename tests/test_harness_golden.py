"""Selection helpers of the validation harness against golden values produced by the reference's own
``metric2objective`` / ``convergence_range_from_finals`` / ``Diffusion.get_best_ids*`` (tests/golden/make_golden.py g8) on
synthetic simulator metrics.  Host-side code: runs without a GPU."""
import json
import os

import numpy as np
import pytest

from dgdm_amd.dynamics import metrics
from tests.golden.make_golden_names import OBJ16

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_harness.json")))


def synth_metrics(seed, n_ori=360):      # same generator as tests/golden/make_golden.py::synth_metrics (inputs only)
    rs = np.random.RandomState(seed)
    walk = np.cumsum(rs.normal(0, 2.0, n_ori))
    return {
        "profile": rs.randint(0, 3, n_ori).astype(np.int64),
        "profile_x": rs.randint(0, 3, n_ori).astype(np.int64),
        "profile_y": rs.randint(0, 3, n_ori).astype(np.int64),
        "delta_theta": rs.normal(0, 0.3, n_ori),
        "final_delta_theta": rs.normal(0, 0.5, n_ori),
        "delta_pos": rs.normal(0, 0.01, (n_ori, 2)),
        "final_pos": rs.normal(0, 0.02, (n_ori, 2)),
        "final_theta": np.where(rs.rand(n_ori) < 0.1, rs.uniform(-180, 180, n_ori), walk),
    }


def _diffusion():
    from dgdm_amd.generator.diffusion import Diffusion
    return object.__new__(Diffusion)          # the helpers use no instance state


def test_metric2objective_and_selection_match_the_reference():
    ms = [synth_metrics(s) for s in GOLD["seeds"]]
    G, O = GOLD["num_grippers"], GOLD["num_objects"]
    d = _diffusion()
    assert len(GOLD["objectives"]) == 2 * (len(OBJ16) + 1)
    for key, want in GOLD["objectives"].items():
        name, lo, hi = key.split("|")
        a, b = int((float(lo) + 1) * 180), int((float(hi) + 1) * 180)
        sliced = [{k: m[k][a:b] for k in m} for m in ms]
        objs = [metrics.metric2objective(m, 'rotate' if name == 'rotate_in_place' else name) for m in sliced]
        assert list(objs[0].keys()) == want["keys"], key
        for got, ref in zip(objs, want["values"]):
            for k in ref:
                assert float(got[k]) == ref[k], (key, k)
        best = d.get_best_ids(objs, G, O, opt_obj=name)
        assert [{k: int(v) for k, v in b_.items()} for b_ in best] == want["best_ids"], key
        avg = [{k: float(np.mean([objs[i * G + g][k] for i in range(O)])) for k in objs[0]} for g in range(G)]
        assert int(d.get_average_best_ids(avg, opt_obj=name)) == want["average_best"], key


def test_convergence_ranges_match_the_reference():
    for case in GOLD["convergence_ranges"]:
        got = metrics.convergence_range_from_finals(np.asarray(case["finals"]), threshold=case["thr"])
        assert [[int(a), int(b)] for a, b in got] == case["ranges"]


def test_error_behaviour():
    assert GOLD["errors"] == {"shift": "NotImplementedError", "rotate_in_place": "NotImplementedError", "selector": "opt obj not supported"}
    m = synth_metrics(1)
    for bad in ("shift", "rotate_in_place", "clockwise", "shift_sideways"):
        with pytest.raises(NotImplementedError):
            metrics.metric2objective(m, bad)
    with pytest.raises(ValueError, match="opt obj not supported"):
        _diffusion().get_average_best_ids([{}], opt_obj="shift")
    with pytest.raises(ValueError, match="opt obj not supported"):
        _diffusion().get_best_ids_all_metrics([{}], opt_obj="spin")
