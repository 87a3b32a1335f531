"""Noise-prediction network with the reference's interface (generator/diffusion_utils.py:25-285).

The module tree below exists to carry the parameters under the reference's ``state_dict`` names
(``down_modules.0.0.blocks.0.block.0.weight`` ...); ``forward`` is one HIP launch (csrc/unet.hip)."""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from .. import engine
from ..dynamics._backed import HipBacked


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        f = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        a = x[:, None] * f[None, :]
        return torch.cat((a.sin(), a.cos()), dim=-1)


class Downsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)


class Upsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)


class Conv1dBlock(nn.Module):
    """Conv1d -> GroupNorm -> Mish (parameters only)."""

    def __init__(self, inp_channels, out_channels, kernel_size, n_groups=8):
        super().__init__()
        self.block = nn.Sequential(nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2),
                                   nn.GroupNorm(n_groups, out_channels), nn.Mish())


class ConditionalResidualBlock1D(nn.Module):
    """Two Conv1dBlocks, FiLM from the step embedding, 1x1 residual when the widths differ (parameters only)."""

    def __init__(self, in_channels, out_channels, cond_dim, kernel_size=3, n_groups=8):
        super().__init__()
        self.out_channels = out_channels
        self.blocks = nn.ModuleList([Conv1dBlock(c, out_channels, kernel_size, n_groups) for c in (in_channels, out_channels)])
        self.cond_encoder = nn.Sequential(nn.Mish(), nn.Linear(cond_dim, 2 * out_channels), nn.Unflatten(-1, (-1, 1)))
        self.residual_conv = nn.Conv1d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()


class ConditionalUnet1D(HipBacked):
    def __init__(self, input_dim: int, global_cond_dim: int, down_dims: List[int], diffusion_step_embed_dim: int,
                 kernel_size: int = 5, n_groups: int = 8):
        super().__init__()
        if input_dim != 1 or global_cond_dim != 0:
            raise NotImplementedError("the HIP U-Net is built for input_dim=1, global_cond_dim=0 (generator/train.py:80)")
        self.input_dim, self.down_dims, self.dsed = input_dim, list(down_dims), diffusion_step_embed_dim
        self.kernel_size, self.n_groups = kernel_size, n_groups
        dsed, cond = diffusion_step_embed_dim, diffusion_step_embed_dim + global_cond_dim
        widths = [input_dim] + list(down_dims)
        levels = list(zip(widths[:-1], widths[1:]))

        def res(a, b):
            return ConditionalResidualBlock1D(a, b, cond_dim=cond, kernel_size=kernel_size, n_groups=n_groups)

        self.diffusion_step_encoder = nn.Sequential(SinusoidalPosEmb(dsed), nn.Linear(dsed, 4 * dsed), nn.Mish(), nn.Linear(4 * dsed, dsed))
        self.mid_modules = nn.ModuleList([res(widths[-1], widths[-1]) for _ in range(2)])
        self.down_modules = nn.ModuleList(
            nn.ModuleList([res(a, b), res(b, b), Downsample1d(b) if i < len(levels) - 1 else nn.Identity()])
            for i, (a, b) in enumerate(levels))
        self.up_modules = nn.ModuleList(
            nn.ModuleList([res(2 * b, a), res(a, a), Upsample1d(a)]) for a, b in reversed(levels[1:]))
        self.final_conv = nn.Sequential(Conv1dBlock(widths[1], widths[1], kernel_size=kernel_size), nn.Conv1d(widths[1], input_dim, 1))

    def _build_handle(self):
        return engine.Unet1d(self.plain_state_dict(), self.down_dims, self.dsed, self.kernel_size, self.n_groups)

    def forward(self, sample: torch.Tensor, timestep: torch.Tensor, global_cond: Optional[torch.Tensor] = None):
        """sample (B, num_points, input_dim), timestep (B,) or 0-dim -> (B, num_points, input_dim)."""
        if global_cond is not None:
            raise NotImplementedError("global_cond_dim is 0 on the reference's path (generator/train.py:80)")
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1)
        return self.handle().forward(sample, t.expand(sample.shape[0]) if t.numel() == 1 else t)
