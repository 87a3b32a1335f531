"""Host-side harness artefacts: per-step PNG dumps and the objective tables the reference sends to
wandb, written as JSON.  The table arithmetic is replayed against golden values produced by the reference's own functions
(tests/golden/g8_harness.json: metric2objective / get_best_ids / get_average_best_ids on synthetic simulator metrics)."""
import json
import os

import numpy as np
import pytest

from dgdm_amd.generator import artefacts
from dgdm_amd.generator.diffusion import Diffusion
from tests.golden.make_golden_names import OBJ16


def synth_metrics(seed, n_ori=360):
    """Same synthetic simulator output as tests/golden/make_golden.py::synth_metrics (inputs only)."""
    rs = np.random.RandomState(seed)
    walk = np.cumsum(rs.normal(0, 2.0, n_ori))
    return {
        "profile": rs.randint(0, 3, n_ori).astype(np.int64), "profile_x": rs.randint(0, 3, n_ori).astype(np.int64),
        "profile_y": rs.randint(0, 3, n_ori).astype(np.int64), "delta_theta": rs.normal(0, 0.3, n_ori),
        "final_delta_theta": rs.normal(0, 0.5, n_ori), "delta_pos": rs.normal(0, 0.01, (n_ori, 2)),
        "final_pos": rs.normal(0, 0.02, (n_ori, 2)),
        "final_theta": np.where(rs.rand(n_ori) < 0.1, rs.uniform(-180, 180, n_ori), walk),
    }


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_harness.json")) as f:
        return json.load(f)


def test_plots_have_the_reference_names(tmp_path):
    v2, v3 = np.linspace(-1, 1, 14), np.linspace(-1, 1, 42)
    for path, v, mode, stacked in ((tmp_path / "val_vis" / "0_3.png", v2, 'point', True), (tmp_path / "val_vis" / "0_4.png", v3, 'point_3d', True),
                                   (tmp_path / "val_vis_noise" / "0_1_4.png", v3, 'point_3d', False)):
        out = artefacts.plot_fingers(str(path), v, mode, 7, 3, stacked=stacked)
        assert os.path.getsize(out) > 1000 and open(out, "rb").read(4) == b"\x89PNG"


def test_unguided_table_matches_reference_selection(tmp_path, golden):
    no, ng = golden["num_objects"], golden["num_grippers"]
    metrics = [synth_metrics(s) for s in golden["seeds"]]
    n = no * ng
    sim_out = ([f"g{i}.png" for i in range(n)], metrics, [f"p{i}" for i in range(n)], [f"px{i}" for i in range(n)],
               [f"py{i}" for i in range(n)], [f"f{i}" for i in range(n)], [[] for _ in range(n)], [f"d{i}" for i in range(n)])
    model = object.__new__(Diffusion)            # the selection helpers use no instance state
    log = artefacts.TableLog(str(tmp_path))
    imgs = [f"last{i}.png" for i in range(ng)]
    for rng in ([-1.0, 1.0], [-0.5, 0.25]):
        for name in OBJ16:
            g = golden["objectives"][f"{name}|{rng[0]}|{rng[1]}"]
            artefacts.unguided_table(model, log, sim_out, imgs, no, ng, name, rng, fingers_3d=False)
            f = tmp_path / "tables" / ("val__unguided_sample__%s_orirange=%.3f_%.3f.json" % (name, rng[0], rng[1]))
            t = json.load(open(f))
            assert t["columns"] == ["object_idx", "gripper_idx", "gripper", "objective", "profile", "profile_x", "profile_y", "final"]
            rows = t["data"]
            assert len(rows) == 3 + n
            for i in range(n):
                assert rows[3 + i][0] == i // ng and rows[3 + i][1] == i % ng and rows[3 + i][2] == imgs[i % ng]
                for k in g["keys"]:
                    assert rows[3 + i][3][k] == pytest.approx(g["values"][i][k], rel=1e-6, abs=1e-9)
            for k in g["keys"]:
                assert rows[0][3][k] == pytest.approx(np.mean([v[k] for v in g["values"]]), rel=1e-6, abs=1e-9)
                best = [g["values"][b[k]][k] for b in g["best_ids"]]
                assert rows[1][3][k] == pytest.approx(np.mean(best), rel=1e-6, abs=1e-9)
            assert rows[2][1] == g["average_best"] and rows[2][2] == imgs[g["average_best"]]


def test_guided_and_multi_object_tables(tmp_path, golden):
    ng = 5
    model = object.__new__(Diffusion)
    log = artefacts.TableLog(str(tmp_path))

    def sim(seed0, n):
        return ([f"g{i}" for i in range(n)], [synth_metrics(seed0 + i) for i in range(n)], [f"p{i}" for i in range(n)], [f"px{i}" for i in range(n)],
                [f"py{i}" for i in range(n)], [f"f{i}" for i in range(n)], [[f"v{i}"] for i in range(n)], [f"d{i}" for i in range(n)])
    # per-object guided table: object i simulated with its ng grippers
    sims = [sim(100 + ng * i, ng) for i in range(3)]
    avg = artefacts.guided_table(model, log, sims, 'shift_left', [-1.0, 1.0])
    t = json.load(open(tmp_path / "tables" / "val__guided_sample__shift_left_orirange=-1.000_1.000.json"))
    g = golden["objectives"]["shift_left|-1.0|1.0"]
    for k, v in avg.items():      # per object the best gripper's own score k, averaged over the objects (:592)
        assert v == pytest.approx(np.mean([g["values"][b[k]][k] for b in g["best_ids"]]), rel=1e-6, abs=1e-9)
    assert t["data"][0][0] == -1 and all(r[3].startswith("py") for r in t["data"][1:])          # shift_left shows profile_y (:588-589)
    with pytest.raises(ValueError, match='opt obj not supported'):
        artefacts.guided_table(model, log, sims, 'wiggle', [-1.0, 1.0])
    # multi-object table: gripper g simulated on all 3 objects; a roll-out that lost an object is dropped (:683-684)
    per_gripper = [sim(100 + g_, 3) for g_ in range(ng)] + [sim(900, 2)]
    best = artefacts.multi_object_table(model, log, per_gripper, 3, 'rotate', [-1.0, 1.0])
    t = json.load(open(tmp_path / "tables" / "val__guided_sample__allobj_rotate_orirange=-1.000_1.000.json"))
    assert t["columns"] == ["gripper", "objective", "last_img", "gripper_dir"] and len(t["data"]) == len(best)
    assert artefacts.multi_object_table(model, log, [sim(900, 2)], 3, 'rotate', [-1.0, 1.0]) is None
