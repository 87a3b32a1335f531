"""Bit-exact index parity on the GPU: the HIP farthest-point sampling and ball queries against the
index arrays the reference's own ``farthest_point_sample`` / ``query_ball_point`` produced (tests/golden/g4_pointnet.npz, incl. the
cloud with exact duplicate points) and against the CPU oracle on re-ordered clouds.  ``np.array_equal`` everywhere."""
import numpy as np
import pytest
import torch

from dgdm_amd import engine, synth
from oracle import dgdm_oracle as orc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dgdm_amd import _lib
    _lib.device_init(0)
    return torch.device("cuda:0")


def _pad_first(lists, counts):
    """query_ball_point pads a short group with its first member (pointnet2_utils.py:112-114)."""
    k = lists.shape[1]
    return np.where(np.arange(k)[None, :] < counts[:, None], lists, lists[:, :1])


def test_fps_and_ball_indices_golden(dev):
    g = util.load("g4_pointnet.npz")
    dyn = engine.Dynamics(3, util.dyn3d_sd(g["seed"]), 42)
    for ci in range(g["clouds"].shape[0]):
        idx = engine.debug_pointnet_indices(dyn, torch.from_numpy(g["clouds"][ci]).to(dev))
        st = int(g["fps_start"][ci])
        assert np.array_equal(idx["fps512"][st], g["fps512"][ci]), ci
        assert np.array_equal(idx["fps128"][st], g["fps128"][ci]), ci
        centres = g["fps128"][ci]
        assert np.array_equal(idx["ball1"][centres], g["ball_r02_n32"][ci]), ci
        assert np.array_equal(_pad_first(idx["ball2"][centres], idx["ball2_count"][centres]), g["ball_r04_n64"][ci]), ci
        # crowded flag = "the r=0.4 ball holds more than 64 points"
        d = orc.square_distance(torch.from_numpy(g["clouds"][ci])[None], torch.from_numpy(g["clouds"][ci])[None])[0]
        assert np.array_equal(idx["crowded"], (~(d > 0.4 ** 2)).sum(1).numpy() > 64), ci


def test_fps_every_start_vs_oracle(dev):
    """All 512 start indices of one clean and one duplicate-point cloud: the FPS tables the product path reads."""
    dyn = engine.Dynamics(3, util.dyn3d_sd(33), 42)
    dup = synth.synth_object_3d(32).clone()
    dup[9] = dup[400]
    dup[10] = dup[400]
    for cloud in (synth.synth_object_3d(31), dup):
        idx = engine.debug_pointnet_indices(dyn, cloud.to(dev))
        starts = torch.arange(512)
        rep = cloud[None].expand(512, -1, -1)
        assert np.array_equal(idx["fps512"], orc.farthest_point_sample(rep, 512, starts).numpy())
        assert np.array_equal(idx["fps128"], orc.farthest_point_sample(rep, 128, starts).numpy())


def test_ball_query_on_reordered_cloud_vs_oracle(dev):
    """sa2's query scans the cloud in sa1's FPS order (pointnet2_utils.py:132-134 on new_xyz of sa1): for a few variants s1 the
    first-64 lists of every centre must equal the oracle's query on the re-ordered cloud, mapped back to point ids."""
    dyn = engine.Dynamics(3, util.dyn3d_sd(33), 42)
    cloud = synth.synth_object_3d(8)          # 146 of its 512 centres have more than 64 points in their r = 0.4 ball
    base = engine.debug_pointnet_indices(dyn, cloud.to(dev))
    assert base["crowded"].sum() > 0, "the case is only interesting when some query truncates at 64"
    for s1 in (0, 77, 511):
        perm = torch.from_numpy(base["fps512"][s1].astype(np.int64))
        idx = engine.debug_pointnet_indices(dyn, cloud.to(dev), perm)
        re = cloud[perm]
        ref = orc.query_ball_point(0.4, 64, re[None], re[None])[0]                 # positions in the re-ordered cloud, centre = position
        ref_ids = perm[ref].numpy()                                                # -> point ids; row j = centre point perm[j]
        got = _pad_first(idx["ball2"], idx["ball2_count"])[perm.numpy()]
        assert np.array_equal(got, ref_ids), s1
