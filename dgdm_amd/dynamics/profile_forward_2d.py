"""2-D dynamics model with the reference's interface (dynamics/profile_forward_2d.py:5-156) on the HIP path."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import engine
from ._backed import HipBacked


class Embedder:
    """x -> [x, sin(f x), cos(f x) for f in freq_bands]   (profile_forward_2d.py:5-38)."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d, n = kwargs['input_dims'], kwargs['num_freqs']
        top = kwargs['max_freq_log2']
        self.freq_bands = 2.0 ** torch.linspace(0.0, top, steps=n) if kwargs['log_sampling'] \
            else torch.linspace(1.0, 2.0 ** top, steps=n)
        self.out_dim = d * ((1 if kwargs['include_input'] else 0) + n * len(kwargs['periodic_fns']))

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        cols = [inputs] if self.kwargs['include_input'] else []
        for f in self.freq_bands:
            cols.extend(fn(inputs * f) for fn in self.kwargs['periodic_fns'])
        return torch.cat(cols, -1)


def get_embedder(input_dims, multires, i=0, scalar_factor=1):
    if i == -1:
        return nn.Identity(), 3
    e = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires,
                 log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return (lambda x, e=e: e.embed(x / scalar_factor)), e.out_dim


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """[cos(t f) | sin(t f)], f_i = max_period^(-i/half)   (profile_forward_2d.py:58-76)."""
    half = dim // 2
    f = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
    a = timesteps[:, None].float() * f[None]
    e = torch.cat([a.cos(), a.sin()], dim=-1)
    return torch.cat([e, torch.zeros_like(e[:, :1])], dim=-1) if dim % 2 else e


def _mlp2(n_in: int, width: int, act: nn.Module) -> nn.Sequential:
    return nn.Sequential(nn.Linear(n_in, width), act, nn.Linear(width, width))


def _trunk(n_in: int, widths) -> nn.Sequential:
    layers = []
    for w in widths:
        layers += [nn.Linear(n_in, w), nn.BatchNorm1d(w), nn.ReLU()]
        n_in = w
    return nn.Sequential(*layers)


class ProfileForward2DModel(HipBacked):
    def __init__(self, W=256, params_ch=400, ori_ch=1, pos_ch=2, output_ch=3, object_ch=20):
        super().__init__()
        if W != 256 or output_ch != 3 or ori_ch != 1 or pos_ch != 2:
            raise NotImplementedError("the HIP trunk is built for W=256, output_ch=3, ori_ch=1, pos_ch=2 (generator/train.py:88)")
        self.W, self.params_ch, self.output_ch, self.object_ch = W, params_ch, output_ch, object_ch
        self.ori_embed, self.ori_ch = get_embedder(ori_ch, 4, 0, scalar_factor=1)
        self.pos_embed, self.pos_ch = get_embedder(pos_ch, 4, 0, scalar_factor=1)
        self.pose_embed_dim = self.ori_ch + self.pos_ch
        self.time_embed_dim = self.object_encode_dim = self.gripper_encode_dim = W
        self.time_encoder = _mlp2(W // 2, W, nn.SiLU())
        self.object_encoder = _mlp2(object_ch, W, nn.ReLU())
        self.gripper_encoder = _mlp2(params_ch, W, nn.ReLU())
        self.linears = _trunk(3 * W + self.pose_embed_dim, [W] * 8)
        self.output = nn.Linear(W, output_ch)

    def _build_handle(self):
        return engine.Dynamics(2, self.plain_state_dict(), self.params_ch, self.object_ch)

    def forward(self, x_ctrl, x_ori, x_pos, timesteps, object_vertices):
        """ctrlpts [rows, params_ch], ori [rows,1], pos [rows,2], timesteps [rows], object [rows, object_ch] -> [rows,3]."""
        return self.handle().forward2d(x_ctrl, x_ori, x_pos, timesteps, object_vertices)
