"""PointNet++ set-abstraction building blocks with the reference's names (dynamics/models/pointnet2_utils.py:27-210).

``square_distance``, ``index_points``, ``farthest_point_sample`` and ``query_ball_point`` run as HIP kernels behind the C-ABI
(``dgdm_square_distance`` / ``dgdm_index_points`` / ``dgdm_farthest_point_sample`` / ``dgdm_query_ball_point``: the same device code
the guided path's table pipeline is built from) and return what the reference returns - int64 indices, the reference's float32
operation order, ``torch.randint`` from the CPU generator for the FPS start (:83).  ``sample_and_group{,_all}`` compose them.
The set-abstraction LAYERS are parameter holders: the reference never calls them on their own (dynamics/models/pointnet2.py:28-30)
and the SSG encoder is evaluated as a whole by ``PointNet2.forward`` (csrc/pointnet.hip explains why not layer by layer)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ..._lib import check, dptr, lib, stream_ptr


def _f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("dgdm_amd runs on an MI355X through libdgdm_hip.so; this tensor is on the CPU and there is no CPU path")
    return t.detach().to(torch.float32).contiguous()


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """src [B,S,3], dst [B,N,3] -> [B,S,N]: -2 src.dst + |src|^2 + |dst|^2 in the reference's operation order (pointnet2_utils.py:27-48)."""
    a, b = _f32(src), _f32(dst)
    B, S, _ = a.shape
    N = b.shape[1]
    out = torch.empty((B, S, N), dtype=torch.float32, device=a.device)
    check(lib().dgdm_square_distance(dptr(a), dptr(b), B, S, N, dptr(out), stream_ptr()))
    return out


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """points [B,N,C], idx [B,S] or [B,S,K] -> gathered [B,S,(K,)C]  (pointnet2_utils.py:51-68)."""
    p = _f32(points)
    B, N, Cc = p.shape
    flat = idx.reshape(B, -1).to(device=p.device, dtype=torch.int32).contiguous()
    out = torch.empty((B, flat.shape[1], Cc), dtype=torch.float32, device=p.device)
    check(lib().dgdm_index_points(dptr(p), dptr(flat), B, N, flat.shape[1], Cc, dptr(out), stream_ptr()))
    return out.reshape(*idx.shape, Cc)


def farthest_point_sample(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """xyz [B,N,3] -> centroid indices [B,npoint] int64; random start per cloud from the CPU generator (pointnet2_utils.py:71-92)."""
    x = _f32(xyz)
    B, N, _ = x.shape
    start = np.ascontiguousarray(torch.randint(0, N, (B,), dtype=torch.long).numpy())
    out = torch.empty((B, npoint), dtype=torch.int32, device=x.device)
    check(lib().dgdm_farthest_point_sample(dptr(x), start.ctypes.data, B, N, int(npoint), dptr(out), stream_ptr()))
    return out.to(torch.int64)


def query_ball_point(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """xyz [B,N,3], new_xyz [B,S,3] -> [B,S,nsample] int64: the first `nsample` in-radius indices, padded with the first (:95-115)."""
    x, c = _f32(xyz), _f32(new_xyz)
    B, N, _ = x.shape
    S = c.shape[1]
    out = torch.empty((B, S, nsample), dtype=torch.int32, device=x.device)
    r2 = float(np.float32(radius ** 2))          # `sqrdists > radius ** 2`: a Python double compared with float32 distances as float32
    check(lib().dgdm_query_ball_point(r2, int(nsample), dptr(x), dptr(c), B, N, S, dptr(out), stream_ptr()))
    return out.to(torch.int64)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """pointnet2_utils.py:118-146: FPS centres, their ball groups, coordinates relative to the centre (+ the grouped features)."""
    B, N, Cc = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    grouped_xyz_norm = grouped_xyz - new_xyz.view(B, npoint, 1, Cc)
    new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1) if points is not None else grouped_xyz_norm
    return (new_xyz, new_points, grouped_xyz, fps_idx) if returnfps else (new_xyz, new_points)


def sample_and_group_all(xyz, points):
    """pointnet2_utils.py:149-166: one group holding every point."""
    B, N, Cc = xyz.shape
    new_xyz = torch.zeros(B, 1, Cc, device=xyz.device)
    grouped_xyz = xyz.view(B, 1, N, Cc)
    new_points = torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1) if points is not None else grouped_xyz
    return new_xyz, new_points


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
        """pointnet2_utils.py:184-210 on the device: xyz [B, C, N], points [B, D, N] or None -> (new_xyz [B, C, S], new_points [B, D', S]).
        Eval mode (BatchNorm running statistics, folded into the 1x1 convolutions in float64); in training mode the layer only exists
        inside the Trainer's fused training step (csrc/train3d.hip), which owns the batch statistics and the gradients.  The guided path
        does not come through here: it evaluates the three levels as per-object tables (PointNet2.forward, csrc/pointnet.hip)."""
        if self.training:
            raise NotImplementedError("training-mode set abstraction runs inside dynamics.trainer.Trainer.step (csrc/train3d.hip); call .eval() for a forward pass")
        xyz = xyz.permute(0, 2, 1)
        pts = points.permute(0, 2, 1) if points is not None else None
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, pts)
        else:
            new_xyz, new_points = sample_and_group(self.npoint, self.radius, self.nsample, xyz, pts)
        B, S, ns, Cin = new_points.shape                                   # [B, npoint, nsample, C + D]
        x = _f32(new_points).reshape(B * S * ns, Cin)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            w = conv.weight.detach().double().reshape(conv.out_channels, -1).cpu()
            sc = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
            wt = (sc[:, None] * w).t().contiguous().float().to(x.device)                        # [Cin][Cout]
            b = (sc * (conv.bias.detach().double().cpu() - bn.running_mean.detach().double().cpu()) + bn.bias.detach().double().cpu()).float().to(x.device)
            y = torch.empty((x.shape[0], conv.out_channels), dtype=torch.float32, device=x.device)
            check(lib().dgdm_linear_act(dptr(x), x.shape[1], dptr(wt), dptr(b), dptr(y), y.shape[1], x.shape[0], x.shape[1], conv.out_channels, 1, stream_ptr()))
            x = y
        out = torch.empty((B * S, x.shape[1]), dtype=torch.float32, device=x.device)
        check(lib().dgdm_group_max(dptr(x), B * S, ns, x.shape[1], dptr(out), stream_ptr()))
        return new_xyz.permute(0, 2, 1), out.reshape(B, S, -1).permute(0, 2, 1)
