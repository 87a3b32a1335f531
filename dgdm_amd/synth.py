"""Deterministic synthetic checkpoints and inputs for the guided-sampling path.

There is no network on the build/GPU boxes, so neither the authors' released
checkpoints nor Icons-50 / scanned objects are available.  Everything the tests,
``bench.py`` and ``smoke()`` feed to the path is generated here from NumPy seeds:

* ``unet_spec`` / ``dyn2d_spec`` / ``dyn3d_spec`` list ``(state_dict key, shape)``
  for the three networks of the path.  The key names are the reference's
  (``generator/diffusion_utils.py:123-236``, ``dynamics/profile_forward_2d.py:78-135``,
  ``dynamics/profile_forward_3d.py:13-65``, ``dynamics/models/pointnet2.py:11-19``) so a
  real checkpoint and a synthetic one are interchangeable; ``tests/golden/make_golden.py``
  asserts the specs against the reference modules' own ``state_dict()``.
* ``synth_state_dict`` fills a spec in sorted-key order from ``RandomState(seed)``
  (He-style scales so eight ReLU layers keep O(1) signal, BatchNorm
  running statistics randomised so eval-mode BN is not the identity).
* ``synth_object_2d`` / ``synth_object_3d`` / ``synth_noise`` build the object point sets
  and the start noise exactly in the form ``generator/train.py:94-124`` and
  ``generator/diffusion.py:182-183`` hand them to the sampler.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

Spec = List[Tuple[str, Tuple[int, ...]]]


# --------------------------------------------------------------------------- specs
def _lin(spec: Spec, name: str, cin: int, cout: int) -> None:
    spec.append((name + ".weight", (cout, cin)))
    spec.append((name + ".bias", (cout,)))


def _bn(spec: Spec, name: str, c: int) -> None:
    for leaf, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)),
                      ("running_var", (c,)), ("num_batches_tracked", ())):
        spec.append((f"{name}.{leaf}", shp))


def _conv_block(spec: Spec, name: str, cin: int, cout: int, k: int) -> None:
    # Conv1d -> GroupNorm -> Mish  (diffusion_utils.py:57-72); Mish holds no tensors
    spec.append((f"{name}.block.0.weight", (cout, cin, k)))
    spec.append((f"{name}.block.0.bias", (cout,)))
    spec.append((f"{name}.block.1.weight", (cout,)))
    spec.append((f"{name}.block.1.bias", (cout,)))


def _res_block(spec: Spec, name: str, cin: int, cout: int, cond: int, k: int) -> None:
    # diffusion_utils.py:75-99
    _conv_block(spec, f"{name}.blocks.0", cin, cout, k)
    _conv_block(spec, f"{name}.blocks.1", cout, cout, k)
    _lin(spec, f"{name}.cond_encoder.1", cond, 2 * cout)
    if cin != cout:
        spec.append((f"{name}.residual_conv.weight", (cout, cin, 1)))
        spec.append((f"{name}.residual_conv.bias", (cout,)))


def unet_spec(input_dim: int = 1, down_dims: Sequence[int] = (128, 256), dsed: int = 32,
              kernel_size: int = 5, global_cond_dim: int = 0) -> Spec:
    """Key/shape list of ``ConditionalUnet1D`` (diffusion_utils.py:144-236)."""
    spec: Spec = []
    dims = [input_dim] + list(down_dims)
    cond = dsed + global_cond_dim
    _lin(spec, "diffusion_step_encoder.1", dsed, 4 * dsed)
    _lin(spec, "diffusion_step_encoder.3", 4 * dsed, dsed)
    pairs = list(zip(dims[:-1], dims[1:]))
    mid = dims[-1]
    for i in range(2):
        _res_block(spec, f"mid_modules.{i}", mid, mid, cond, kernel_size)
    for lvl, (ci, co) in enumerate(pairs):
        _res_block(spec, f"down_modules.{lvl}.0", ci, co, cond, kernel_size)
        _res_block(spec, f"down_modules.{lvl}.1", co, co, cond, kernel_size)
        if lvl < len(pairs) - 1:
            spec.append((f"down_modules.{lvl}.2.conv.weight", (co, co, 3)))
            spec.append((f"down_modules.{lvl}.2.conv.bias", (co,)))
    for lvl, (ci, co) in enumerate(reversed(pairs[1:])):
        _res_block(spec, f"up_modules.{lvl}.0", 2 * co, ci, cond, kernel_size)
        _res_block(spec, f"up_modules.{lvl}.1", ci, ci, cond, kernel_size)
        # `is_last = ind >= len(in_out) - 1` is never true for the up path
        # (diffusion_utils.py:205), so every up level carries a ConvTranspose1d.
        spec.append((f"up_modules.{lvl}.2.conv.weight", (ci, ci, 4)))
        spec.append((f"up_modules.{lvl}.2.conv.bias", (ci,)))
    _conv_block(spec, "final_conv.0", dims[1], dims[1], kernel_size)
    spec.append(("final_conv.1.weight", (input_dim, dims[1], 1)))
    spec.append(("final_conv.1.bias", (input_dim,)))
    return spec


def _trunk(spec: Spec, first_in: int, widths: Sequence[int], out_ch: int) -> None:
    cin = first_in
    for i, w in enumerate(widths):
        _lin(spec, f"linears.{3 * i}", cin, w)
        _bn(spec, f"linears.{3 * i + 1}", w)
        cin = w
    _lin(spec, "output", cin, out_ch)


def dyn2d_spec(params_ch: int = 14, object_ch: int = 200, W: int = 256, output_ch: int = 3) -> Spec:
    """``ProfileForward2DModel`` (profile_forward_2d.py:78-135). pose embedding = 9 + 18."""
    spec: Spec = []
    _lin(spec, "time_encoder.0", W // 2, W)
    _lin(spec, "time_encoder.2", W, W)
    _lin(spec, "object_encoder.0", object_ch, W)
    _lin(spec, "object_encoder.2", W, W)
    _lin(spec, "gripper_encoder.0", params_ch, W)
    _lin(spec, "gripper_encoder.2", W, W)
    _trunk(spec, 3 * W + 27, [W] * 8, output_ch)
    return spec


def pointnet2_spec(prefix: str, out_ch: int = 256) -> Spec:
    """``PointNet2`` SSG encoder (pointnet2.py:17-19, pointnet2_utils.py:175-181)."""
    spec: Spec = []
    for sa, cin, mlp in (("sa1", 3, (64, 128)), ("sa2", 131, (128, out_ch)), ("sa3", out_ch + 3, (out_ch,))):
        c = cin
        for i, co in enumerate(mlp):
            spec.append((f"{prefix}{sa}.mlp_convs.{i}.weight", (co, c, 1, 1)))
            spec.append((f"{prefix}{sa}.mlp_convs.{i}.bias", (co,)))
            _bn(spec, f"{prefix}{sa}.mlp_bns.{i}", co)
            c = co
    return spec


def dyn3d_spec(params_ch: int = 42, W: int = 256, output_ch: int = 3) -> Spec:
    """``ProfileForward3DModel`` (profile_forward_3d.py:13-65); ``time_encoder`` exists but is unused."""
    spec: Spec = []
    _lin(spec, "time_encoder.0", W // 2, W)
    _lin(spec, "time_encoder.2", W, W)
    spec += pointnet2_spec("object_encoder.", W)
    _lin(spec, "gripper_encoder.0", params_ch, W)
    _lin(spec, "gripper_encoder.2", W, W)
    _trunk(spec, 3 * W + 27, [2 * W] + [W] * 7, output_ch)
    return spec


# --------------------------------------------------------------------------- weights
def synth_state_dict(spec: Spec, seed: int, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Fill ``spec`` deterministically (sorted key order, one RandomState)."""
    rs = np.random.RandomState(seed)
    out: Dict[str, torch.Tensor] = {}
    for key, shape in sorted(spec):
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        is_norm = ".mlp_bns." in key or (key.startswith("linears.") and len(shape) == 1 and _is_bn_key(key)) \
            or ".block.1." in key
        if leaf == "running_mean":
            v = rs.normal(0.0, 0.1, size=shape)
        elif leaf == "running_var":
            v = rs.uniform(0.5, 1.5, size=shape)
        elif is_norm and leaf == "weight":
            v = rs.uniform(0.5, 1.5, size=shape)
        elif is_norm and leaf == "bias":
            v = rs.normal(0.0, 0.1, size=shape)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if "up_modules" in key and key.endswith(".2.conv.weight"):
                fan_in = shape[0] * shape[2] // 2   # ConvTranspose1d: (cin, cout, k), stride 2
            bound = gain * math.sqrt(6.0 / fan_in)  # He-uniform: keeps ReLU/Mish stacks O(1)
            v = rs.uniform(-bound, bound, size=shape)
        else:  # bias of a conv / linear
            v = rs.uniform(-0.1, 0.1, size=shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))
    return out


def scale_output(sd: Dict[str, torch.Tensor], gain: float) -> Dict[str, torch.Tensor]:
    """Copy of a dynamics state_dict with the output layer scaled by `gain`.

    He-init trunks put out guidance gradients 10^2-10^4 times eps' size (summed coherently over the C = G*P^2 cells), whereas the
    reference's classifier scales (0.001 / 0.5, generator/diffusion.py:30-33) are tuned for a trained model whose guidance term is
    of eps' order.  With gain 1 the guided chain is a chaotic map that the reference does not reproduce across its own CPU thread
    counts (tests/golden/g9_*: 0.2 .. 4.5 finger L2); the full-grid parity fixtures therefore also use calibrated gains."""
    out = dict(sd)
    out["output.weight"] = sd["output.weight"] * gain
    out["output.bias"] = sd["output.bias"] * gain
    return out


def _is_bn_key(key: str) -> bool:
    # linears.<3i+1>.* are BatchNorm1d, linears.<3i>.* are Linear
    idx = int(key.split(".")[1])
    return idx % 3 == 1


# --------------------------------------------------------------------------- inputs
def synth_noise(seed: int, batch: int, num_points: int, input_dim: int = 1) -> torch.Tensor:
    """Start noise exactly as ``generator/diffusion.py:182-183`` draws it."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.randn(batch, num_points, input_dim)).float()


def synth_object_2d(seed: int, num_vertices: int = 100) -> torch.Tensor:
    """Closed star-shaped contour inside +-0.05 m, normalised as ``generator/train.py:111-124``."""
    rs = np.random.RandomState(10_000 + seed)
    ang = np.linspace(0.0, 2.0 * np.pi, num_vertices, endpoint=False)
    rad = 0.03 + 0.012 * rs.uniform(-1, 1) * np.cos(2 * ang + rs.uniform(0, 6.28)) \
        + 0.008 * rs.uniform(-1, 1) * np.cos(3 * ang + rs.uniform(0, 6.28)) \
        + 0.004 * rs.uniform(-1, 1) * np.cos(5 * ang + rs.uniform(0, 6.28))
    pts = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)
    pts = (pts + 0.05) / 0.1 * 2.0 - 1.0
    return torch.from_numpy(pts.astype(np.float32))


def synth_object_3d(seed: int, num_vertices: int = 512) -> torch.Tensor:
    """Points sampled uniformly on the surface of a random box or ellipsoid that fits
    x,y in +-0.1 m, z in [0, 0.12] m, normalised as ``generator/train.py:94-109``."""
    rs = np.random.RandomState(20_000 + seed)
    half = np.array([rs.uniform(0.03, 0.09), rs.uniform(0.03, 0.09), rs.uniform(0.02, 0.06)])
    if seed % 2 == 0:      # ellipsoid (area-weighting ignored: density only needs to be surface-like)
        v = rs.normal(size=(num_vertices, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        pts = v * half
    else:                  # box: choose a face by area, then a point on it
        areas = np.array([half[1] * half[2], half[0] * half[2], half[0] * half[1]])
        face = rs.choice(3, size=num_vertices, p=areas / areas.sum())
        sign = rs.choice([-1.0, 1.0], size=num_vertices)
        pts = rs.uniform(-1, 1, size=(num_vertices, 3)) * half
        pts[np.arange(num_vertices), face] = sign * half[face]
    pts[:, 2] += half[2]   # rest on z = 0
    lo = np.array([-0.1, -0.1, 0.0])
    hi = np.array([0.1, 0.1, 0.12])
    pts = (pts - lo) / (hi - lo) * 2.0 - 1.0
    return torch.from_numpy(pts.astype(np.float32))


OBJECTIVES_12 = ['convergence', 'shift_up', 'shift_down', 'shift_left', 'shift_right', 'rotate_clockwise',
                 'rotate_counterclockwise', 'rotate', 'clockwise_up', 'clockwise_left', 'counterclockwise_up',
                 'counterclockwise_left']   # generator/diffusion.py:307
