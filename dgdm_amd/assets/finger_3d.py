"""3-D finger surfaces from control points (reference: assets/finger_3d.py:60-98), on the MI355X.

``generate_3d_ctrlpts`` / ``generate_3d_finger_vertices`` / ``generate_3d_gripper`` keep the reference's names and return
values; ``generate_3d_grippers`` is the batched form for the sampler's output (dynamics/sim_test_mj_3d.py:233-237)."""
from __future__ import annotations

import numpy as np
import torch

from .. import engine


def _net():
    # x = linspace(-0.12, 0.12, 7), z = linspace(0, 0.12, 3), control point (i, j) -> index 3 i + j  (finger_3d.py:77-80)
    x = np.linspace(-0.12, 0.12, 7)
    z = np.linspace(0, 0.12, 3)
    x_n, z_n = np.meshgrid(x, z)
    return x_n.T.reshape(-1), z_n.T.reshape(-1)


def generate_3d_ctrlpts(yl, yr):
    xs, zs = _net()
    return np.concatenate((np.stack([xs, np.asarray(yl, dtype=np.float64), zs], -1),
                           np.stack([xs, np.asarray(yr, dtype=np.float64), zs], -1)), axis=0)


def generate_3d_grippers(samples: torch.Tensor, sample_size: int = 25) -> torch.Tensor:
    """Sampler output (B, 42, 1) in [-1, 1] -> (B, 2, sample_size^2, 3): both finger surfaces in metres."""
    return engine.finger_decode_3d(samples, sample_size)


def generate_3d_finger_vertices(control_points, degree_u=3, degree_v=2, sample_size=25):
    """One finger: (21, 3) control points on the reference's net -> (sample_size^2, 3) surface points."""
    if (degree_u, degree_v) != (3, 2):
        raise NotImplementedError("the device decode is built for the reference's degrees (3, 2)")
    cp = np.asarray(control_points, dtype=np.float64).reshape(21, 3)
    xs, zs = _net()
    if not (np.allclose(cp[:, 0], xs, rtol=0, atol=1e-9) and np.allclose(cp[:, 2], zs, rtol=0, atol=1e-9)):
        raise NotImplementedError("device decode supports the reference's 7 x 3 control net (finger_3d.py:77-80)")
    y = np.concatenate([cp[:, 1], cp[:, 1]]).astype(np.float32)
    dev = torch.device("cuda", torch.cuda.current_device())
    return engine.finger_decode_3d(torch.from_numpy(y).reshape(1, -1).to(dev), int(sample_size), scale=1.0, offset=0.0)[0, 0].cpu().numpy().astype(np.float64)


def generate_3d_gripper(yl, yr, sample_size=25):
    """Returns (ctrlpts (42, 3), vertices (2 sample_size^2, 3)) like the reference."""
    y = np.concatenate([np.asarray(yl, dtype=np.float32), np.asarray(yr, dtype=np.float32)])
    dev = torch.device("cuda", torch.cuda.current_device())
    v = engine.finger_decode_3d(torch.from_numpy(y).reshape(1, -1).to(dev), int(sample_size), scale=1.0, offset=0.0)[0].cpu().numpy()
    return generate_3d_ctrlpts(yl, yr), np.concatenate([v[0], v[1]], 0).astype(np.float64)
