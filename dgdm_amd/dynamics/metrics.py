"""Host-side helpers of the 'convergence' objective (reference: dynamics/metrics.py:4-38).

These run on B-length / G-length integer profiles on the host, exactly where the reference runs them
(generator/diffusion.py:532-538 builds the profile with a Python loop on the CPU).  ``metric2objective`` and
``convergence_range_from_finals`` of the reference score MuJoCo roll-outs and are outside the path.
"""
from __future__ import annotations

import numpy as np
import torch


def slicer(a, lower, upper):
    """Wrap-around window a[lower:upper] with Python slice clamping (metrics.py:32-38)."""
    lower, upper, n = int(lower), int(upper), len(a)
    if lower < 0:
        pieces = (a[lower:], a[:upper])
    elif upper > n:
        pieces = (a[lower:], a[:upper - n])
    else:
        return a[lower:upper]
    return torch.cat(pieces) if isinstance(a, torch.Tensor) else np.concatenate(pieces)


def convergence_mode(profile: torch.Tensor):
    """Runs of ones followed by runs of zeros on the circular binary profile (metrics.py:4-21).

    Returns (lengths, points): for every 1->0 transition inside the first period, its index (the last 1)
    and the distance between the surrounding 0->1 transitions measured on the twice-unrolled profile."""
    dev = profile.device
    bits = (profile.detach().cpu().numpy() > 0).astype(np.int64)
    n = bits.size
    ones = int(bits.sum())
    if ones == 0:
        return torch.tensor([n], device=dev), torch.tensor([0], device=dev)
    if ones == n:
        return torch.tensor([n], device=dev), torch.tensor([n - 1], device=dev)
    step = np.diff(np.concatenate([bits, bits]))
    falls = np.nonzero(step < 0)[0]
    falls = falls[falls < n]
    rises = np.nonzero(step > 0)[0]
    marks = np.concatenate([[0], rises[rises > falls[0]], [2 * n]])
    lengths = np.diff(marks)[:falls.size]
    return torch.from_numpy(lengths).to(dev), torch.from_numpy(falls).to(dev)


def convergence_mode_three_class(profile: torch.Tensor):
    """Classes {0: clockwise, 1: static, 2: counter-clockwise}; static cells are dropped first (metrics.py:23-30)."""
    keep = torch.nonzero(profile != 1).reshape(-1)
    if keep.numel() == 0:
        z = torch.tensor([0], device=profile.device)
        return z, z.clone()
    lengths, points = convergence_mode(profile[keep])
    return lengths, keep[points]


def metric2objective(*args, **kwargs):
    raise NotImplementedError("metric2objective scores MuJoCo roll-outs (reference dynamics/metrics.py:67-234); "
                              "simulation is outside the MI355X guided-sampling path (SURVEY.md §2 #10)")
