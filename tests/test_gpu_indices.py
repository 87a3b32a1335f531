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


def test_reference_named_index_functions(dev):
    """dynamics.models.pointnet2_utils with the reference's names and signatures, HIP-backed: bit-exact against the reference's golden
    index arrays and against the oracle on clouds of another size (N = 300, npoint = 64, radius 0.3, nsample = 16)."""
    from dynamics.models import pointnet2_utils as pu
    g = util.load("g4_pointnet.npz")
    clouds = torch.from_numpy(g["clouds"]).to(dev)
    st = torch.from_numpy(g["fps_start"].astype(np.int64))
    orig = torch.randint
    torch.randint = lambda *a, **k: st.clone()                      # the draw the golden arrays were made with
    try:
        fps512, fps128 = pu.farthest_point_sample(clouds, 512), pu.farthest_point_sample(clouds, 128)
    finally:
        torch.randint = orig
    assert fps512.dtype == torch.int64 and np.array_equal(fps512.cpu().numpy(), g["fps512"]) and np.array_equal(fps128.cpu().numpy(), g["fps128"])
    new_xyz = pu.index_points(clouds, fps128)
    assert torch.equal(new_xyz.cpu(), torch.from_numpy(g["clouds"])[torch.arange(4)[:, None], torch.from_numpy(g["fps128"]).long()])
    assert np.array_equal(pu.query_ball_point(0.2, 32, clouds, new_xyz).cpu().numpy(), g["ball_r02_n32"])
    assert np.array_equal(pu.query_ball_point(0.4, 64, clouds, new_xyz).cpu().numpy(), g["ball_r04_n64"])
    # other sizes, against the oracle (the CPU restatement of the same functions)
    rs = np.random.RandomState(3)
    xyz = torch.from_numpy(rs.uniform(-1, 1, (5, 300, 3)).astype(np.float32))
    torch.manual_seed(9)
    fps = pu.farthest_point_sample(xyz.to(dev), 64).cpu()
    torch.manual_seed(9)
    assert torch.equal(fps, orc.farthest_point_sample(xyz, 64, torch.randint(0, 300, (5,), dtype=torch.long)))
    cen = xyz[torch.arange(5)[:, None], fps]
    assert torch.equal(pu.query_ball_point(0.3, 16, xyz.to(dev), cen.to(dev)).cpu(), orc.query_ball_point(0.3, 16, xyz, cen))
    # float values of the expanded form: the CPU BLAS's K = 3 dot order is its own affair (fma or not), so a few float32 ulps of |x|^2 ~ 3
    assert float((pu.square_distance(cen.to(dev), xyz.to(dev)).cpu() - orc.square_distance(cen, xyz)).abs().max()) < 2e-6
    # an empty ball: every entry is N, as the reference's masked assignment leaves it
    far = torch.full((5, 1, 3), 9.0)
    assert torch.equal(pu.query_ball_point(0.3, 16, xyz.to(dev), far.to(dev)).cpu(), orc.query_ball_point(0.3, 16, xyz, far))
    # sample_and_group: shapes and the relative coordinates of the reference
    torch.manual_seed(10)
    nx, npts = pu.sample_and_group(32, 0.5, 8, xyz.to(dev), None)
    assert nx.shape == (5, 32, 3) and npts.shape == (5, 32, 8, 3) and float(npts.norm(dim=-1).max()) <= 0.5 + 1e-6
