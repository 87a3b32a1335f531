"""2-D finger curves from control points (reference: assets/finger_sampler.py:39-51 ``generate_gripper``), on the MI355X.

``generate_gripper`` keeps the reference's name, arguments and return values for one gripper; ``generate_grippers`` is the
batched form the sampler's output goes through (dynamics/sim_test_mj.py:254-262 does the same per gripper on the host)."""
from __future__ import annotations

import numpy as np
import torch

from .. import engine


def generate_grippers(samples: torch.Tensor, num_points: int = 200) -> torch.Tensor:
    """Sampler output (B, L, 1) in [-1, 1] -> (B, 2, num_points, 2): the left and right finger curves in metres."""
    return engine.finger_decode_2d(samples, num_points)


def generate_gripper(finger_x, finger_yl, finger_yr, num_points):
    """Returns (ctrlpts (2K, 2), allpts (2 num_points, 2)) like the reference.  ``finger_x`` must be the abscissae the reference
    always passes, ``np.linspace(-0.12, 0.12, K)`` (sim_test_mj.py:257): the device decode has that grid built in."""
    finger_x = np.asarray(finger_x, dtype=np.float64)
    K = finger_x.shape[0]
    if not np.allclose(finger_x, np.linspace(-0.12, 0.12, K), rtol=0, atol=1e-9):
        raise NotImplementedError("device decode supports finger_x = linspace(-0.12, 0.12, K) (the reference's only call site)")
    y = np.concatenate([np.asarray(finger_yl, dtype=np.float32), np.asarray(finger_yr, dtype=np.float32)])
    dev = torch.device("cuda", torch.cuda.current_device())
    pts = engine.finger_decode_2d(torch.from_numpy(y).reshape(1, -1).to(dev), int(num_points), scale=1.0, offset=0.0)[0].cpu().numpy()
    ctrl = np.concatenate([np.stack([finger_x, np.asarray(finger_yl, dtype=np.float64)], -1),
                           np.stack([finger_x, np.asarray(finger_yr, dtype=np.float64)], -1)], 0)
    return ctrl, np.concatenate([pts[0], pts[1]], 0).astype(np.float64)
