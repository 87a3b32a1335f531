"""PointNet++ set-abstraction building blocks with the reference's names
(dynamics/models/pointnet2_utils.py:27-210).  The index functions run as HIP kernels; the
set-abstraction layers are parameter holders - the SSG encoder is evaluated as a whole by
``PointNet2.forward`` (see csrc/pointnet.hip for why it is not evaluated layer by layer)."""
from __future__ import annotations

import torch
import torch.nn as nn


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """points [B,N,C], idx [B,S] or [B,S,K] -> gathered [B,S,(K,)C]  (pointnet2_utils.py:51-68)."""
    B = points.shape[0]
    flat = idx.reshape(B, -1)
    out = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1]))
    return out.reshape(*idx.shape, points.shape[-1])


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """Expanded-form pairwise squared distance, in the reference's operation order (pointnet2_utils.py:27-48)."""
    d = -2 * torch.matmul(src, dst.transpose(1, 2))
    d = d + (src ** 2).sum(-1)[:, :, None]
    return d + (dst ** 2).sum(-1)[:, None, :]


class PointNetSetAbstraction(nn.Module):
    """Holds mlp_convs / mlp_bns exactly as the reference registers them (pointnet2_utils.py:169-182)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        for c_out in mlp:
            self.mlp_convs.append(nn.Conv2d(in_channel, c_out, 1))
            self.mlp_bns.append(nn.BatchNorm2d(c_out))
            in_channel = c_out

    def forward(self, xyz, points):
        raise NotImplementedError("set-abstraction layers are evaluated fused, through PointNet2.forward (the reference "
                                  "never calls them on their own: dynamics/models/pointnet2.py:28-30)")
