"""CPU oracle for the finger-geometry decode.  TEST INFRASTRUCTURE ONLY (see oracle/dgdm_oracle.py).

2-D follows assets/finger_sampler.py:39-51 with the scaling of dynamics/sim_test_mj.py:257-262 and uses the reference's own
dependency, ``scipy.interpolate.CubicSpline`` (present in this image): that half is pinned by construction.

3-D follows assets/finger_3d.py:60-81 with the scaling of dynamics/sim_test_mj_3d.py:236-237.  The reference evaluates the surface
with ``geomdl`` (requirements.txt: geomdl==5.3.1), which is absent here: **parity unpinned** against geomdl itself.  The surface
is restated from the published definition - B-spline basis by the Cox-de Boor recursion, tensor product, geomdl's documented
clamped uniform knot vector ``[0]*p + linspace(0, 1, n - p + 1) + [1]*p``, evaluation parameters ``linspace(0, 1, sample_size)``
in u-major order - and anchored on an independent implementation, ``scipy.interpolate.BSpline`` (tests/test_decode.py)."""
from __future__ import annotations

import numpy as np
from scipy.interpolate import CubicSpline


def decode_2d(samples: np.ndarray, num_points: int = 200) -> np.ndarray:
    """samples (B, L) in [-1, 1] -> (B, 2, num_points, 2) metres."""
    samples = np.asarray(samples, dtype=np.float64)
    B, L = samples.shape
    K = L // 2
    out = np.empty((B, 2, num_points, 2))
    p_x = np.linspace(-0.12, 0.12, K)                       # sim_test_mj.py:257
    for b in range(B):
        p_y = samples[b] * 0.03 - 0.015                      # :261
        for f in range(2):
            cs = CubicSpline(p_x, p_y[f * K:(f + 1) * K])    # finger_sampler.py:40,43
            x_new = np.linspace(p_x.min(), p_x.max(), num_points)
            out[b, f, :, 0] = x_new
            out[b, f, :, 1] = cs(x_new)
    return out


def knot_vector(degree: int, n: int) -> np.ndarray:
    return np.concatenate([np.zeros(degree), np.linspace(0.0, 1.0, n - degree + 1), np.ones(degree)])


def basis(degree: int, kv: np.ndarray, n: int, u: float) -> np.ndarray:
    """N_i,degree(u), i < n (Cox-de Boor; the last knot belongs to the last non-empty span)."""
    m = len(kv)
    N = np.zeros(m - 1)
    span = n - 1
    for i in range(m - 1):
        if kv[i] <= u < kv[i + 1]:
            span = i
    N[span] = 1.0
    for d in range(1, degree + 1):
        for i in range(m - 1 - d):
            a = (u - kv[i]) / (kv[i + d] - kv[i]) * N[i] if kv[i + d] > kv[i] else 0.0
            b = (kv[i + d + 1] - u) / (kv[i + d + 1] - kv[i + 1]) * N[i + 1] if kv[i + d + 1] > kv[i + 1] else 0.0
            N[i] = a + b
    return N[:n]


def control_net(y21: np.ndarray) -> np.ndarray:
    x = np.linspace(-0.12, 0.12, 7)                          # finger_3d.py:77-80
    z = np.linspace(0, 0.12, 3)
    x_n, z_n = np.meshgrid(x, z)
    return np.stack([x_n.T.reshape(-1), y21, z_n.T.reshape(-1)], axis=-1).reshape(7, 3, 3)


def surface(ctrl: np.ndarray, sample_size: int) -> np.ndarray:
    """ctrl (7, 3, 3) -> (sample_size^2, 3), u-major."""
    ku, kv = knot_vector(3, 7), knot_vector(2, 3)
    us = np.linspace(0.0, 1.0, sample_size)
    Nu = np.stack([basis(3, ku, 7, u) for u in us])          # (S, 7)
    Nv = np.stack([basis(2, kv, 3, v) for v in us])          # (S, 3)
    return np.einsum('ai,bj,ijc->abc', Nu, Nv, ctrl).reshape(-1, 3)


def decode_3d(samples: np.ndarray, sample_size: int = 25) -> np.ndarray:
    """samples (B, 42) in [-1, 1] -> (B, 2, sample_size^2, 3) metres."""
    samples = np.asarray(samples, dtype=np.float64)
    B = samples.shape[0]
    out = np.empty((B, 2, sample_size * sample_size, 3))
    for b in range(B):
        p_y = samples[b].reshape(-1) * 0.05 - 0.05           # sim_test_mj_3d.py:236-237
        for f in range(2):
            out[b, f] = surface(control_net(p_y[f * 21:(f + 1) * 21]), sample_size)
    return out
