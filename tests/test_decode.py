"""Finger-geometry decode: oracle vs independent scipy implementations on the CPU, HIP vs oracle on
the GPU.  Tolerance: 2e-7 m absolute (float32 evaluation of a 7- or 21-term sum of values <= 0.12 m)."""
import numpy as np
import pytest
import torch
from scipy.interpolate import BSpline

from oracle import finger_decode_oracle as dec


def test_oracle_surface_matches_scipy_bspline():
    rs = np.random.RandomState(0)
    ku, kv = dec.knot_vector(3, 7), dec.knot_vector(2, 3)
    assert np.allclose(ku, [0, 0, 0, 0, .25, .5, .75, 1, 1, 1, 1]) and np.allclose(kv, [0, 0, 0, 1, 1, 1])
    ctrl = dec.control_net(rs.uniform(-0.1, 0.0, 21))
    S = 9
    got = dec.surface(ctrl, S).reshape(S, S, 3)
    us = np.linspace(0, 1, S)
    # tensor product with scipy's B-splines: first along u for every (j, coordinate), then along v
    along_u = BSpline(ku, ctrl.reshape(7, 9), 3)(us).reshape(S, 3, 3)              # (S, j, c)
    want = np.stack([BSpline(kv, along_u[a], 2)(us) for a in range(S)])            # (S, S, c)
    assert np.abs(got - want).max() < 1e-14
    # corners interpolate the corner control points (clamped knots), rows of the basis sum to one
    assert np.allclose(got[0, 0], ctrl[0, 0]) and np.allclose(got[-1, -1], ctrl[-1, -1])
    assert np.allclose(sum(dec.basis(3, ku, 7, 0.3)), 1.0) and np.allclose(sum(dec.basis(2, kv, 3, 1.0)), 1.0)


def test_oracle_2d_scaling_and_interpolation():
    rs = np.random.RandomState(1)
    s = rs.uniform(-1, 1, (3, 14))
    out = dec.decode_2d(s, 50)
    assert out.shape == (3, 2, 50, 2)
    x = np.linspace(-0.12, 0.12, 7)
    # the spline passes through the scaled control points; x runs over [-0.12, 0.12]
    for f in range(2):
        for k in range(7):
            j = np.argmin(np.abs(out[1, f, :, 0] - x[k]))
            if abs(out[1, f, j, 0] - x[k]) < 1e-12:
                assert abs(out[1, f, j, 1] - (s[1, f * 7 + k] * 0.03 - 0.015)) < 1e-12
    assert out[..., 0].min() == -0.12 and out[..., 0].max() == 0.12
    assert out[..., 1].min() > -0.08 and out[..., 1].max() < 0.05


@pytest.mark.gpu
def test_decode_hip_vs_oracle():
    from dgdm_amd import _lib, engine
    from dgdm_amd.assets import finger_3d, finger_sampler
    _lib.device_init(0)
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(2)
    for B, n in ((1, 200), (5, 100), (64, 7), (3, 1)):
        s = rs.uniform(-1, 1, (B, 14)).astype(np.float32)
        got = engine.finger_decode_2d(torch.from_numpy(s).reshape(B, 14, 1).to(dev), n).cpu().numpy()
        assert np.abs(got - dec.decode_2d(s, n)).max() < 2e-7, (B, n)
    for B, S in ((1, 25), (4, 25), (32, 10), (2, 1)):
        s = rs.uniform(-1, 1, (B, 42)).astype(np.float32)
        got = engine.finger_decode_3d(torch.from_numpy(s).reshape(B, 42, 1).to(dev), S).cpu().numpy()
        assert np.abs(got - dec.decode_3d(s, S)).max() < 2e-7, (B, S)
    # reference-named entry points, one gripper, physical units
    x = np.linspace(-0.12, 0.12, 7)
    yl, yr = rs.uniform(-0.045, 0.015, 7), rs.uniform(-0.045, 0.015, 7)
    ctrl, allpts = finger_sampler.generate_gripper(x, yl, yr, 100)
    assert ctrl.shape == (14, 2) and allpts.shape == (200, 2)
    want = dec.decode_2d(((np.concatenate([yl, yr]) + 0.015) / 0.03)[None], 100)[0]
    assert np.abs(allpts - np.concatenate([want[0], want[1]])).max() < 2e-7
    y3l, y3r = rs.uniform(-0.1, 0.0, 21), rs.uniform(-0.1, 0.0, 21)
    c3, v3 = finger_3d.generate_3d_gripper(y3l, y3r, sample_size=25)
    assert c3.shape == (42, 3) and v3.shape == (1250, 3)
    assert np.allclose(c3, finger_3d.generate_3d_ctrlpts(y3l, y3r))
    w3 = dec.decode_3d(((np.concatenate([y3l, y3r]) + 0.05) / 0.05)[None], 25)[0]
    assert np.abs(v3 - np.concatenate([w3[0], w3[1]])).max() < 2e-7
    one = finger_3d.generate_3d_finger_vertices(c3[:21].tolist(), sample_size=25)
    assert np.abs(one - w3[0]).max() < 2e-7
    with pytest.raises(NotImplementedError):
        finger_sampler.generate_gripper(np.linspace(-0.1, 0.1, 7), yl, yr, 10)
    with pytest.raises(_lib.DgdmError):
        engine.finger_decode_3d(torch.zeros(2, 40, device=dev))
