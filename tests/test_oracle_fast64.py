"""oracle/fast64.py (the per-object float64 yardstick of the 3-D chains) against dgdm_oracle's float64 mode - the float64 evaluation of the
as-written dataflow row by row.  CPU only."""
import os

import numpy as np
import torch

from dgdm_amd import synth
from oracle import dgdm_oracle as orc
from oracle import fast64
from tests import util

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_embeddings_match_rowwise_float64_pointnet():
    """Per-object tables -> a row's embedding, against PointNet++ evaluated in float64 on the row's replicated cloud with the same two
    FPS start draws (dynamics/models/pointnet2.py:21-32).  Draws chosen to hit crowded and uncrowded sa2 centres."""
    sd64 = fast64._f64(util.dyn3d_sd(0))
    xyz = synth.synth_object_3d(50)
    s1 = torch.tensor([0, 3, 3, 117, 400, 511])
    s2 = torch.tensor([5, 0, 127, 64, 90, 1])
    tab = fast64.ObjectTables64(sd64, xyz, s1_only=sorted(set(s1.tolist())))
    assert int(tab.crowded.sum()) > 0 and int((~tab.crowded).sum()) > 0
    got = tab.embed(s1, s2)
    cloud = xyz.double().t()[None].expand(s1.numel(), -1, -1).contiguous()
    want = orc.pointnet2_forward(sd64, cloud, orc.StartLog([s1, s2]), prefix="object_encoder.")
    assert got.shape == want.shape == (6, 256)
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-13, err


def test_recorded_chains_match_the_rowwise_float64_chains():
    """The float64 chains fast64 wrote into g9_calls64.npz against the ones dgdm_oracle's float64 mode wrote into g9_f64.npz (25-50 CPU
    minutes each, round 3; tests/golden/make_golden.py g9_f64 / g9_calls64): the same chain end points to float64 rounding."""
    slow, fast = np.load(os.path.join(GOLD, "g9_f64.npz")), np.load(os.path.join(GOLD, "g9_calls64.npz"))
    parts = [k[3:-6] for k in slow.files if k.startswith("3d/") and k.endswith("_chain")]
    assert len(parts) == 6
    for p in parts:
        a, b = slow[f"3d/{p}_chain"], fast[f"{p}/chain"]
        assert a.shape == b.shape
        assert float(np.abs(a - b).max()) < 1e-10 * max(1.0, float(np.abs(a).max())), p
