"""CPU oracle for the DGDM guided-sampling hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Nothing under ``dgdm_amd/`` does: the product path is the HIP
library and fails loudly when it is missing.

What this is: a plain PyTorch-CPU fp32 restatement, in functional form over flat
``state_dict``s carrying the reference's key names, of every function on the path
(SURVEY.md §8(a)).  It deliberately keeps the reference's *as-written* dataflow - the
R-row replicated inputs of ``cond_fn``, the ``sub_bs`` loop with gradient accumulation,
PointNet++ evaluated on every replica, ``torch.autograd.grad`` for the guidance gradient,
the CPU-generator ``torch.randint`` FPS starts - so that it is an independent check of the
de-duplicated HIP dataflow.  Each function cites the reference lines it follows
(paths relative to /root/reference).

Pinning: the reference has no tests or golden vectors.  The model
arithmetic (a4-a12) is pinned by ``tests/golden/*.npz``, produced here by
``tests/golden/make_golden.py`` from the reference's own modules (``generator/diffusion.py``
imported through third-party stubs) and compared with this file in
``tests/test_oracle_golden.py``.  The DDIM scheduler (a13) lives in un-vendored
``diffusers==0.11.1`` (requirements.txt:1) which is absent from this image: that part is
restated from the published algorithm and is **parity unpinned**; it is anchored only on
the reference's call sites and the closed forms / probe values recorded in SURVEY.md §8(a13,c).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# generator/diffusion.py:30-33
SCALE_2D, SCALE_2D_CONV, SCALE_3D, SCALE_3D_CONV = 0.001, 10.0, 0.5, 0.8


# =========================================================================== a13  DDIM
class DDIM:
    """diffusers 0.11.1 ``DDIMScheduler`` as the reference constructs it
    (generator/train.py:83: squaredcos_cap_v2, clip_sample=True, epsilon prediction;
    ``step`` is called with the default eta=0 at generator/diffusion.py:201,256,576,647)."""

    def __init__(self, num_train_timesteps: int):
        T = num_train_timesteps

        def abar(s: float) -> float:
            return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2

        betas = [min(1.0 - abar((i + 1) / T) / abar(i / T), 0.999) for i in range(T)]  # python float64
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)      # set_alpha_to_one=True (default)
        self.num_train_timesteps = T
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, T)[::-1].copy().astype(np.int64))

    def set_timesteps(self, n: int) -> None:
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

    def add_noise(self, x0: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        a = self.alphas_cumprod[timesteps]
        sa = (a ** 0.5).flatten()
        sb = ((1 - a) ** 0.5).flatten()
        while sa.dim() < x0.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * x0 + sb * noise

    def step(self, eps: torch.Tensor, t: int, x: torch.Tensor) -> torch.Tensor:
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        x0 = torch.clamp(x0, -1, 1)                         # clip_sample=True
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps     # eta = 0: no variance term


# =========================================================================== a7  U-Net
def _rc(x: torch.Tensor, w: torch.Tensor):
    """bf16-contraction mode (see `contraction` below): a convolution whose input AND output have more than one channel runs on
    the matrix pipe with its input and weight rounded to bf16 (float32 accumulation, float32 bias); the single-channel
    input/output convolutions, the Linear layers, GroupNorm, Mish, FiLM and the residual adds stay float32."""
    if _CONTRACTION == 'bf16' and x.shape[1] > 1 and w.shape[0] > 1 and w.shape[1] > 1:
        return _bf(x), _bf(w)
    return x, w


def _conv_gn_mish(sd: SD, p: str, x: torch.Tensor, groups: int = 8) -> torch.Tensor:
    # generator/diffusion_utils.py:57-72
    w = sd[p + ".block.0.weight"]
    xc, wc = _rc(x, w)
    x = F.conv1d(xc, wc, sd[p + ".block.0.bias"], padding=w.shape[-1] // 2)
    x = F.group_norm(x, groups, sd[p + ".block.1.weight"], sd[p + ".block.1.bias"])
    return F.mish(x)


def _film_res_block(sd: SD, p: str, x: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
    # generator/diffusion_utils.py:101-120
    out = _conv_gn_mish(sd, p + ".blocks.0", x)
    e = F.linear(F.mish(cond), sd[p + ".cond_encoder.1.weight"], sd[p + ".cond_encoder.1.bias"])
    c = out.shape[1]
    out = e[:, :c, None] * out + e[:, c:, None]
    out = _conv_gn_mish(sd, p + ".blocks.1", out)
    if p + ".residual_conv.weight" in sd:
        xc, wc = _rc(x, sd[p + ".residual_conv.weight"])
        x = F.conv1d(xc, wc, sd[p + ".residual_conv.bias"])
    return out + x


def unet1d_forward(sd: SD, sample: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
    """``ConditionalUnet1D.forward`` (generator/diffusion_utils.py:238-285); sample (B,L,C), timestep (B,)."""
    x = sample.transpose(1, 2)
    dsed = sd["diffusion_step_encoder.1.weight"].shape[1]
    half = dsed // 2
    fr = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))     # :30-37
    arg = timestep.expand(x.shape[0])[:, None] * fr[None, :]
    g = torch.cat((arg.sin(), arg.cos()), dim=-1).to(sd["diffusion_step_encoder.1.weight"].dtype)      # no-op in float32
    g = F.linear(g, sd["diffusion_step_encoder.1.weight"], sd["diffusion_step_encoder.1.bias"])
    g = F.linear(F.mish(g), sd["diffusion_step_encoder.3.weight"], sd["diffusion_step_encoder.3.bias"])
    n_down = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("down_modules."))
    n_up = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("up_modules."))
    skips: List[torch.Tensor] = []
    for lvl in range(n_down):                                                 # :265-269
        x = _film_res_block(sd, f"down_modules.{lvl}.0", x, g)
        x = _film_res_block(sd, f"down_modules.{lvl}.1", x, g)
        skips.append(x)
        if f"down_modules.{lvl}.2.conv.weight" in sd:
            xc, wc = _rc(x, sd[f"down_modules.{lvl}.2.conv.weight"])
            x = F.conv1d(xc, wc, sd[f"down_modules.{lvl}.2.conv.bias"], stride=2, padding=1)
    for i in range(2):                                                        # :271-272
        x = _film_res_block(sd, f"mid_modules.{i}", x, g)
    for lvl in range(n_up):                                                   # :274-278
        x = torch.cat((x, skips.pop()), dim=1)
        x = _film_res_block(sd, f"up_modules.{lvl}.0", x, g)
        x = _film_res_block(sd, f"up_modules.{lvl}.1", x, g)
        if f"up_modules.{lvl}.2.conv.weight" in sd:
            xc, wc = _rc(x, sd[f"up_modules.{lvl}.2.conv.weight"])
            x = F.conv_transpose1d(xc, wc, sd[f"up_modules.{lvl}.2.conv.bias"], stride=2, padding=1)
    x = _conv_gn_mish(sd, "final_conv.0", x)                                  # :280
    x = F.conv1d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return x.transpose(1, 2)


# =========================================================================== a8  2-D dynamics
def nerf_embed(x: torch.Tensor, nfreq: int = 4) -> torch.Tensor:
    """``get_embedder(d, 4)``: [x, sin(2^k x), cos(2^k x)] k=0..3 (dynamics/profile_forward_2d.py:10-56)."""
    parts = [x]
    for f in 2.0 ** torch.linspace(0.0, nfreq - 1, steps=nfreq):
        parts += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(parts, -1)


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """dynamics/profile_forward_2d.py:58-76 ([cos | sin], frequencies exp(-ln(P) i / half))."""
    half = dim // 2
    fr = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    a = t[:, None].float() * fr[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def _mlp2(sd: SD, p: str, x: torch.Tensor, act: Callable) -> torch.Tensor:
    # the cast is a no-op for a float32 state_dict; with a float64 copy of the weights it makes the whole model run in float64
    # on the same float32 embedding inputs (the triangulation of tests/test_gpu_fullgrid.py: which float32 result is nearer)
    x = act(F.linear(x.to(sd[p + ".0.weight"].dtype), sd[p + ".0.weight"], sd[p + ".0.bias"]))
    return F.linear(x, sd[p + ".2.weight"], sd[p + ".2.bias"])


# --- bf16-contraction mode (BASELINE configs[4]; no counterpart in the reference, which is float32 throughout) -------------
# States on the CPU the rounding points of dgdm_amd/csrc/trunk_bf16.hip so that kernel has something to be checked against:
# BatchNorm folded into the Linear in float64, the folded weight rounded once to bf16; every operand entering a trunk
# contraction (activations forward, gradients backward) rounded to bf16 (nearest even); float32 accumulation, biases,
# first-layer terms that do not depend on the row's object embedding, objective and sums.  Switched on with
# ``with contraction('bf16'):`` around cond_fn / guided_sample*.
_CONTRACTION = 'f32'


class contraction:
    def __init__(self, dtype: str):
        assert dtype in ('f32', 'bf16')
        self.dtype = dtype

    def __enter__(self):
        global _CONTRACTION
        self.prev, _CONTRACTION = _CONTRACTION, self.dtype

    def __exit__(self, *a):
        global _CONTRACTION
        _CONTRACTION = self.prev


def _bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _BfLinear(torch.autograd.Function):
    """y = bf(x) bf(W)^T ;  dL/dx = bf(dL/dy) bf(W)   (float32 accumulation; W is frozen)."""

    @staticmethod
    def forward(ctx, x, w):
        wb = _bf(w)
        ctx.save_for_backward(wb)
        return _bf(x) @ wb.t()

    @staticmethod
    def backward(ctx, g):
        (wb,) = ctx.saved_tensors
        return _bf(g) @ wb, None


def _folded(sd: SD, i: int):
    w, b = sd[f"linears.{3 * i}.weight"].double(), sd[f"linears.{3 * i}.bias"].double()
    n = f"linears.{3 * i + 1}"
    sc = sd[n + ".weight"].double() / torch.sqrt(sd[n + ".running_var"].double() + 1e-5)
    return (sc[:, None] * w).float(), (sc * (b - sd[n + ".running_mean"].double()) + sd[n + ".bias"].double()).float()


def _trunk_bf16(sd: SD, x: torch.Tensor, n_embed: int) -> torch.Tensor:
    """The trunk with bf16 contractions.  ``n_embed`` leading input columns (3-D: the 256 PointNet++ features, which differ
    per row) go through a bf16 contraction in layer 1; the rest of layer 1 is the float32 table part."""
    w, b = _folded(sd, 0)
    z = F.linear(x[:, n_embed:], w[:, n_embed:], b)
    if n_embed:
        z = z + _BfLinear.apply(x[:, :n_embed], w[:, :n_embed])
    a = F.relu(z)
    i = 1
    while f"linears.{3 * i}.weight" in sd:
        w, b = _folded(sd, i)
        a = F.relu(_BfLinear.apply(a, w) + b)
        i += 1
    return _BfLinear.apply(a, sd["output.weight"]) + sd["output.bias"]


TRUNK_CAPTURE: Optional[list] = None     # test hook: when a list, _trunk appends the first layer's pre-activation tensor to it
TRUNK_MARGIN: Optional[list] = None      # test hook: when a list, _trunk appends per row the smallest RELATIVE distance of any ReLU input from zero


def _relu_margin(x_in: torch.Tensor, z: torch.Tensor, sd: SD, i: int) -> torch.Tensor:
    """Per row: min over the units of layer i of |z_j| / (sum_k |w'_jk| |x_k| + |b'_j|), w' / b' the BatchNorm-folded layer - the size of
    the sum whose rounding decides the sign of z_j (a float32 evaluation is off by ~1e-7 .. 1e-6 of it)."""
    with torch.no_grad():
        n = f"linears.{3 * i + 1}"
        sc = sd[n + ".weight"].double() / torch.sqrt(sd[n + ".running_var"].double() + 1e-5)
        w = (sc[:, None] * sd[f"linears.{3 * i}.weight"].double()).abs()
        b = (sc * (sd[f"linears.{3 * i}.bias"].double() - sd[n + ".running_mean"].double()) + sd[n + ".bias"].double()).abs()
        size = x_in.detach().double().abs() @ w.t() + b
        return (z.detach().double().abs() / size.clamp_min(1e-300)).min(dim=1).values


def _trunk(sd: SD, x: torch.Tensor, n_embed: int = 0) -> torch.Tensor:
    # Linear -> BatchNorm1d(eval) -> ReLU x8, then Linear (profile_forward_2d.py:109-135,154-155)
    if _CONTRACTION == 'bf16':
        return _trunk_bf16(sd, x, n_embed)
    i = 0
    margin = None
    while f"linears.{3 * i}.weight" in sd:
        x_in = x
        x = F.linear(x, sd[f"linears.{3 * i}.weight"], sd[f"linears.{3 * i}.bias"])
        b = f"linears.{3 * i + 1}"
        x = F.batch_norm(x, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"],
                         training=False, eps=1e-5)
        if TRUNK_CAPTURE is not None and i == 0:
            TRUNK_CAPTURE.append(x)          # z1: first-layer pre-activation (tests: per-row d objective / d z1, see make_golden.g9_tiles)
        if TRUNK_MARGIN is not None:
            m = _relu_margin(x_in, x, sd, i)
            margin = m if margin is None else torch.minimum(margin, m)
        x = F.relu(x)
        i += 1
    if TRUNK_MARGIN is not None:
        TRUNK_MARGIN.append(margin)
    return F.linear(x, sd["output.weight"], sd["output.bias"])


def dyn2d_forward(sd: SD, x_ctrl, x_ori, x_pos, timesteps, object_vertices) -> torch.Tensor:
    """``ProfileForward2DModel.forward`` (dynamics/profile_forward_2d.py:137-156)."""
    g = _mlp2(sd, "gripper_encoder", x_ctrl, F.relu)
    pose = torch.cat([nerf_embed(x_ori), nerf_embed(x_pos)], dim=1)
    o = _mlp2(sd, "object_encoder", object_vertices, F.relu)
    W = sd["time_encoder.2.weight"].shape[0]
    te = _mlp2(sd, "time_encoder", timestep_embedding(timesteps, W // 2), F.silu)
    return _trunk(sd, torch.cat([o, g, pose, te], dim=1))


# =========================================================================== a10-a12  PointNet++
class StartLog:
    """FPS start indices.  ``draw`` takes them from the torch CPU generator exactly as
    dynamics/models/pointnet2_utils.py:83 does and records them, so that the HIP path can be
    fed the identical indices; ``replay`` hands back previously recorded/forced ones."""

    def __init__(self, forced: Optional[List[torch.Tensor]] = None):
        self.log: List[torch.Tensor] = []
        self._forced = list(forced) if forced is not None else None

    def draw(self, n_points: int, rows: int) -> torch.Tensor:
        if self._forced is not None:
            s = self._forced.pop(0)
            assert s.shape == (rows,)
        else:
            s = torch.randint(0, n_points, (rows,), dtype=torch.long)
        self.log.append(s.clone())
        return s


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    # pointnet2_utils.py:27-48: the expanded form, in this operation order
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d += torch.sum(src ** 2, -1)[:, :, None]
    d += torch.sum(dst ** 2, -1)[:, None, :]
    return d


def farthest_point_sample(xyz: torch.Tensor, npoint: int, start: torch.Tensor) -> torch.Tensor:
    # pointnet2_utils.py:71-92; first-max tie-break comes from torch.max
    B, N, _ = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.long)
    dist = torch.ones(B, N) * 1e10
    far = start
    rows = torch.arange(B)
    for i in range(npoint):
        out[:, i] = far
        c = xyz[rows, far, :].view(B, 1, 3)
        d = torch.sum((xyz - c) ** 2, -1)
        dist = torch.where(d < dist, d, dist)
        far = torch.max(dist, -1)[1]
    return out


def _gather(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    # index_points, pointnet2_utils.py:51-68
    B = points.shape[0]
    bi = torch.arange(B).view([B] + [1] * (idx.dim() - 1)).expand_as(idx)
    return points[bi, idx, :]


def query_ball_point(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    # pointnet2_utils.py:95-115: first `nsample` in-radius indices in ascending order, padded with the first
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.arange(N).view(1, 1, N).repeat(B, S, 1)
    idx[square_distance(new_xyz, xyz) > radius ** 2] = N
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, :1].expand(-1, -1, nsample)
    return torch.where(idx == N, first, idx)


def set_abstraction(sd: SD, p: str, xyz: torch.Tensor, points: Optional[torch.Tensor], npoint, radius, nsample,
                    starts: Optional[StartLog], training: bool = False, buffers: Optional[SD] = None):
    """``PointNetSetAbstraction.forward`` (pointnet2_utils.py:184-210); xyz (B,3,N), points (B,D,N)."""
    xyz = xyz.permute(0, 2, 1)
    if points is not None:
        points = points.permute(0, 2, 1)
    B, N, C = xyz.shape
    if npoint is None:                                    # sample_and_group_all :149-166
        new_xyz = torch.zeros(B, 1, C)
        feat = xyz.view(B, 1, N, C)
        if points is not None:
            feat = torch.cat([feat, points.view(B, 1, N, -1)], dim=-1)
    else:                                                 # sample_and_group :118-146
        # .float() is a no-op on the float32 path; with float64 inputs (the float64 yardstick of tests/test_gpu_fullgrid.py)
        # it keeps the discrete decisions - which points FPS picks, which fall inside a ball - the float32 ones
        fps = farthest_point_sample(xyz.float(), npoint, starts.draw(N, B))
        new_xyz = _gather(xyz, fps)
        idx = query_ball_point(radius, nsample, xyz.float(), new_xyz.float())
        feat = _gather(xyz, idx) - new_xyz.view(B, npoint, 1, C)
        if points is not None:
            feat = torch.cat([feat, _gather(points, idx)], dim=-1)
    feat = feat.permute(0, 3, 2, 1)                       # (B, C+D, nsample, npoint)
    i = 0
    while f"{p}.mlp_convs.{i}.weight" in sd:
        b = f"{p}.mlp_bns.{i}"
        if _CONTRACTION == 'bf16' and npoint is None and i == 0:
            # bf16-contraction mode, sa3 (the 259 -> 256 layer on the 512 sa2 features of every row): BatchNorm folded into the
            # conv in float64; the 256 feature channels go through a bf16 contraction (operands rounded, float32 accumulation),
            # the 3 coordinate channels and the bias stay float32 - the split csrc/pointnet.hip z16_kernel makes.
            w = sd[f"{p}.mlp_convs.{i}.weight"].double().reshape(-1, feat.shape[1])
            sc = sd[b + ".weight"].double() / torch.sqrt(sd[b + ".running_var"].double() + 1e-5)
            wf = (sc[:, None] * w).float()
            bf_ = (sc * (sd[f"{p}.mlp_convs.{i}.bias"].double() - sd[b + ".running_mean"].double()) + sd[b + ".bias"].double()).float()
            rows = feat.permute(0, 2, 3, 1).reshape(-1, feat.shape[1])                     # (B * nsample * npoint, 259)
            out = _BfLinear.apply(rows[:, C:], wf[:, C:]) + F.linear(rows[:, :C], wf[:, :C], bf_)
            feat = F.relu(out).reshape(feat.shape[0], feat.shape[2], feat.shape[3], -1).permute(0, 3, 1, 2)
            i += 1
            continue
        feat = F.conv2d(feat, sd[f"{p}.mlp_convs.{i}.weight"], sd[f"{p}.mlp_convs.{i}.bias"])
        bufs = sd if buffers is None else buffers      # training: batch statistics over (B, nsample, npoint), running statistics updated in place
        feat = F.batch_norm(feat, bufs[b + ".running_mean"], bufs[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"],
                            training=training, momentum=0.1, eps=1e-5)
        feat = F.relu(feat)
        i += 1
    return new_xyz.permute(0, 2, 1), torch.max(feat, 2)[0]


def pointnet2_forward(sd: SD, xyz: torch.Tensor, starts: Optional[StartLog] = None, prefix: str = "", training: bool = False,
                      buffers: Optional[SD] = None) -> torch.Tensor:
    """``PointNet2.forward`` (dynamics/models/pointnet2.py:21-32); xyz (B,3,N) -> (B,256)."""
    starts = starts or StartLog()
    kw = dict(training=training, buffers=buffers)
    l1x, l1p = set_abstraction(sd, prefix + "sa1", xyz, None, 512, 0.2, 32, starts, **kw)
    l2x, l2p = set_abstraction(sd, prefix + "sa2", l1x, l1p, 128, 0.4, 64, starts, **kw)
    _, l3p = set_abstraction(sd, prefix + "sa3", l2x, l2p, None, None, None, None, **kw)
    return l3p.reshape(xyz.shape[0], -1)


def dyn3d_forward(sd: SD, x_ctrl, x_ori, x_pos, timesteps, object_vertices, starts: Optional[StartLog] = None):
    """``ProfileForward3DModel.forward`` (dynamics/profile_forward_3d.py:67-86): channel 1 of x_ctrl only,
    raw 256-d timestep embedding (``time_encoder`` is never called)."""
    g = _mlp2(sd, "gripper_encoder", x_ctrl[:, 1, :], F.relu)
    pose = torch.cat([nerf_embed(x_ori), nerf_embed(x_pos)], dim=1)
    o = pointnet2_forward(sd, object_vertices, starts, prefix="object_encoder.")
    W = sd["gripper_encoder.2.weight"].shape[0]
    te = timestep_embedding(timesteps, W)
    return _trunk(sd, torch.cat([o, g, pose, te], dim=1), n_embed=o.shape[1])


# =========================================================================== a5/a6  objectives
def slicer(a: torch.Tensor, lower: int, upper: int) -> torch.Tensor:
    # dynamics/metrics.py:32-38
    if lower < 0:
        return torch.cat((a[lower:], a[:upper]))
    if upper > len(a):
        return torch.cat((a[lower:], a[:upper - len(a)]))
    return a[lower:upper]


_LINEAR_OBJ = {  # generator/diffusion.py:433-468: coefficient of (dtheta, dx, dy)
    'rotate_clockwise': (-1, 0, 0), 'rotate_counterclockwise': (1, 0, 0), 'shift_up': (0, -1, 0),
    'shift_down': (0, 1, 0), 'shift_left': (0, 0, -1), 'shift_right': (0, 0, 1),
    'clockwise_up': (-1, -1, 0), 'clockwise_down': (-1, 1, 0), 'clockwise_left': (-1, 0, -1),
    'clockwise_right': (-1, 0, 1), 'counterclockwise_up': (1, -1, 0), 'counterclockwise_down': (1, 1, 0),
    'counterclockwise_left': (1, 0, -1), 'counterclockwise_right': (1, 0, 1),
}


def deltas_to_objective(deltas: torch.Tensor, opt_obj: str, centers=None, grid_size: int = 360, num_pos: int = 5):
    """``Diffusion.deltas_to_objective`` (generator/diffusion.py:430-471)."""
    if opt_obj == 'rotate':
        return deltas[..., 0] ** 2
    if opt_obj in _LINEAR_OBJ:
        c = _LINEAR_OBJ[opt_obj]
        out = None
        for j in range(3):
            if c[j]:
                term = deltas[..., j] if c[j] > 0 else -deltas[..., j]
                out = term if out is None else out + term
        return out
    if opt_obj == 'convergence':
        cells = grid_size * num_pos ** 2
        half = (grid_size // 2) * num_pos ** 2
        parts = []
        for i, c in enumerate(centers):
            c = int(c)
            dth = deltas[i * cells:(i + 1) * cells, 0]
            parts.append(slicer(dth, c * num_pos ** 2 - half, c * num_pos ** 2))
            parts.append(slicer(-dth, c * num_pos ** 2, c * num_pos ** 2 + half))
        return torch.cat(parts, dim=0)
    raise ValueError('opt obj not supported')


def convergence_mode(profile: torch.Tensor):
    # dynamics/metrics.py:4-21
    profile = torch.where(profile > 0, 1.0, 0.0)
    n = len(profile)
    if torch.all(profile == 0):
        return torch.tensor([n]), torch.tensor([0])
    if torch.all(profile == 1):
        return torch.tensor([n]), torch.tensor([n - 1])
    profile = torch.cat((profile, profile), dim=0)
    diff = torch.diff(profile)
    conv = torch.where(diff < 0)[0]
    conv = conv[conv < n]
    start = torch.where(diff > 0)[0]
    lengths = torch.diff(torch.cat((torch.tensor([0]), start[start > conv[0]], torch.tensor([2 * n]))))
    return lengths[:len(conv)], conv


def convergence_mode_three_class(profile: torch.Tensor):
    # dynamics/metrics.py:23-30
    ids = torch.where(profile != 1)[0]
    if len(ids) == 0:
        return torch.tensor([0]), torch.tensor([0])
    lengths, pts = convergence_mode(profile[profile != 1])
    return lengths, ids[pts]


# =========================================================================== a4  cond_fn (as written)
class Setup:
    """What ``Diffusion.__init__`` keeps for the guided path (generator/diffusion.py:88-118)."""

    def __init__(self, mode: str, unet: SD, dyn: SD, sched: DDIM, num_points: int, grid_size: int, num_pos: int,
                 sub_batch_size: int = 1024):
        self.mode, self.unet, self.dyn, self.sched = mode, unet, dyn, sched
        self.num_points, self.grid_size, self.num_pos, self.sub_batch_size = num_points, grid_size, num_pos, sub_batch_size
        if mode == 'point_3d':
            thr, std = torch.tensor([0.02, 0.001, 0.001]), torch.tensor([0.0312, 0.0016, 0.0026])
        else:
            thr, std = torch.tensor([0.03, 0.002, 0.003]), torch.tensor([0.0565, 0.0026, 0.0047])
        self.threshold_std = thr / std


def _pose_grid(s: Setup, B: int, ori_range):
    # generator/diffusion.py:478-482: 'ij' meshgrid, every cell value repeated B times -> row r = cell*B + b
    o, px, py = torch.meshgrid(torch.linspace(ori_range[0], ori_range[1], s.grid_size),
                               torch.linspace(-1, 1, s.num_pos), torch.linspace(-1, 1, s.num_pos), indexing='ij')
    ori = o.reshape(-1).repeat_interleave(B).reshape(-1, 1)
    pos = torch.stack([px.reshape(-1).repeat_interleave(B), py.reshape(-1).repeat_interleave(B)], dim=-1)
    return ori, pos


def _pts3d(s: Setup, x: torch.Tensor) -> torch.Tensor:
    # generator/diffusion.py:489: channels [linspace ; x ; linspace]; only channel 1 reaches the model
    B = x.shape[0]
    lin = torch.linspace(-1.0, 1.0, s.num_points // 2).repeat(B * 2, 1).reshape(B, 1, -1)
    return torch.cat([lin, x.transpose(-1, -2), lin], dim=1)


def cond_fn(s: Setup, x: torch.Tensor, t: torch.Tensor, opt_obj: str, object_vertices: torch.Tensor,
            ori_range=(-1.0, 1.0), centers=None, starts: Optional[StartLog] = None) -> torch.Tensor:
    """``Diffusion.cond_fn`` (generator/diffusion.py:473-504)."""
    T = s.sched.num_train_timesteps
    with torch.enable_grad():
        x = x.detach().requires_grad_(True)
        B = x.shape[0]
        cells = s.grid_size * s.num_pos ** 2
        ori, pos = _pose_grid(s, B, ori_range)
        tt = t.repeat(cells).float() / T
        kw = dict(centers=centers, grid_size=s.grid_size, num_pos=s.num_pos)
        if s.mode == 'point':
            pts = x.repeat(cells, 1, 1).reshape(B * cells, -1)
            obj = object_vertices.reshape(1, -1).expand(B * cells, -1)
            logits = dyn2d_forward(s.dyn, pts, ori, pos, tt, obj)
            return torch.autograd.grad(deltas_to_objective(logits, opt_obj, **kw).sum(), x)[0]
        if s.mode == 'point_3d':
            pts = _pts3d(s, x).repeat(cells, 1, 1)
            obj = object_vertices.t().unsqueeze(0)            # (1,3,N), replicated per sub-batch below
            grad = 0.0
            for i in range(0, B * cells, s.sub_batch_size):
                j = min(i + s.sub_batch_size, B * cells)
                logits = dyn3d_forward(s.dyn, pts[i:j], ori[i:j], pos[i:j], tt[i:j], obj.expand(j - i, -1, -1), starts)
                grad = grad + torch.autograd.grad(deltas_to_objective(logits, opt_obj, **kw).sum(), x)[0]
            return grad
        raise ValueError('model type not supported')


def get_convergence_centers(s: Setup, unguided: torch.Tensor, object_vertices: torch.Tensor, ori_range=(-1.0, 1.0),
                            starts: Optional[StartLog] = None) -> torch.Tensor:
    """``Diffusion.get_convergence_centers`` (generator/diffusion.py:506-539)."""
    B = unguided.shape[0]
    G = s.grid_size
    with torch.no_grad():
        ori = torch.linspace(ori_range[0], ori_range[1], G).repeat_interleave(B).reshape(-1, 1)
        pos = torch.zeros(B * G, 2)
        tt = torch.zeros(B * G)
        if s.mode == 'point':
            pts = unguided.repeat(G, 1, 1).reshape(B * G, -1)
            logits = dyn2d_forward(s.dyn, pts, ori, pos, tt, object_vertices.reshape(1, -1).expand(B * G, -1))
        else:
            pts = _pts3d(s, unguided).repeat(G, 1, 1)
            obj = object_vertices.t().unsqueeze(0)
            chunks = []
            for i in range(0, B * G, s.sub_batch_size):
                j = min(i + s.sub_batch_size, B * G)
                chunks.append(dyn3d_forward(s.dyn, pts[i:j], ori[i:j], pos[i:j], tt[i:j], obj.expand(j - i, -1, -1), starts))
            logits = torch.cat(chunks, dim=0)
    th = s.threshold_std[0]
    d0 = logits[..., 0]
    prof = torch.where(d0 > th, 2.0, torch.where(d0 < -th, 0.0, 1.0))                        # :532
    out = []
    for i in range(B):
        lengths, cs = convergence_mode_three_class(prof[torch.arange(i, B * G, B)])
        out.append(cs[torch.argmax(lengths)])
    return torch.stack(out, dim=0)


# =========================================================================== a1-a3  denoise loops
def classifier_scale(mode: str, opt_obj: str, multi: bool = False) -> float:
    # generator/diffusion.py:549-560 and :631-636 (the multi-object loop never uses the *_CONV scales)
    if mode == 'point':
        return SCALE_2D_CONV if (opt_obj == 'convergence' and not multi) else SCALE_2D
    if mode == 'point_3d':
        return SCALE_3D_CONV if (opt_obj == 'convergence' and not multi) else SCALE_3D
    return 0.001


def unguided_sample(s: Setup, x: torch.Tensor) -> torch.Tensor:
    """Loop bodies of generator/diffusion.py:193-201 and :249-256 (bookkeeping dropped)."""
    B = x.shape[0]
    x = x.clone()
    for t in s.sched.timesteps:
        with torch.no_grad():
            eps = unet1d_forward(s.unet, x, t * torch.ones(B, dtype=torch.int64))
        x = s.sched.step(eps, t, x)
    return x


def guided_sample(s: Setup, noise: torch.Tensor, object_vertices: torch.Tensor, opt_obj: str, ori_range=(-1.0, 1.0),
                  unguided: Optional[torch.Tensor] = None, starts: Optional[StartLog] = None,
                  trace: Optional[list] = None) -> torch.Tensor:
    """One object of ``Diffusion.guided_sample`` (generator/diffusion.py:561-576)."""
    B = noise.shape[0]
    centers = get_convergence_centers(s, unguided, object_vertices, ori_range, starts) if opt_obj == 'convergence' else None
    scale = classifier_scale(s.mode, opt_obj)
    x = noise.clone().detach()
    for t in s.sched.timesteps:
        ts = t * torch.ones(B, dtype=torch.int64)
        with torch.no_grad():
            eps = unet1d_forward(s.unet, x, ts)
        g = cond_fn(s, x, ts, opt_obj, object_vertices, ori_range, centers, starts)
        if trace is not None:
            trace.append((eps.clone(), g.clone()))
        eps = eps - (1 - s.sched.alphas_cumprod[t]).sqrt() * g * scale
        x = s.sched.step(eps, t, x)
    return x


def guided_sample_multi_object(s: Setup, noise: torch.Tensor, objects: Sequence[torch.Tensor], opt_obj: str,
                               ori_range=(-1.0, 1.0), starts: Optional[StartLog] = None) -> torch.Tensor:
    """``Diffusion.guided_sample_multi_object`` loop (generator/diffusion.py:637-647)."""
    B = noise.shape[0]
    scale = classifier_scale(s.mode, opt_obj, multi=True)
    x = noise.clone().detach()
    for t in s.sched.timesteps:
        ts = t * torch.ones(B, dtype=torch.int64)
        with torch.no_grad():
            eps = unet1d_forward(s.unet, x, ts)
        g = 0.0
        for ov in objects:
            g = g + cond_fn(s, x, ts, opt_obj, ov, ori_range, None, starts)
        g = g / len(objects)
        eps = eps - (1 - s.sched.alphas_cumprod[t]).sqrt() * g * scale
        x = s.sched.step(eps, t, x)
    return x


# =========================================================================== (f) rank 4  Trainer.step of the 2-D dynamics model
class Trainer2D:
    """``Trainer`` (dynamics/trainer.py:16-106) for ``ProfileForward2DModel``, restated functionally over a flat state_dict:
    ``step`` = trainer.py:53-103 without sub-batches (the 2-D configuration, dynamics/train_dynamics_2d.sh), the model in
    training mode (BatchNorm1d batch statistics, running statistics momentum 0.1 with the unbiased variance), nn.MSELoss,
    torch.autograd for the gradients, and torch.optim.Adam(lr, betas=(0.9, 0.95), weight_decay) (:46) written out
    (single-tensor form of torch 2.x: lerp first moment, bias corrections, eps added after the corrected square root)."""

    def __init__(self, sd: SD, num_train_timesteps: int, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.95), eps: float = 1e-8):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = [k for k, v in self.sd.items() if v.dtype.is_floating_point and "running_" not in k]
        self.ddim = DDIM(num_train_timesteps)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.t = 0
        self.grads: SD = {}
        self.draws = None
        self.relu_margin = float("inf")

    def _forward(self, sd: SD, x_ctrl, x_ori, x_pos, timesteps, object_vertices, training: bool, buffers: Optional[SD] = None) -> torch.Tensor:
        buffers = self.sd if buffers is None else buffers      # BatchNorm running statistics (updated in place when training)
        # profile_forward_2d.py:137-156
        def relu(z):        # every ReLU input passes here: relu_margin = how far the nearest one is from its kink (see tests)
            self.relu_margin = min(self.relu_margin, float(z.detach().abs().min()))
            return F.relu(z)
        g = _mlp2(sd, "gripper_encoder", x_ctrl, relu)
        pose = torch.cat([nerf_embed(x_ori), nerf_embed(x_pos)], dim=1)
        o = _mlp2(sd, "object_encoder", object_vertices, relu)
        te = _mlp2(sd, "time_encoder", timestep_embedding(timesteps, 128), F.silu)
        x = torch.cat([o, g, pose, te], dim=1)
        for i in range(8):
            x = F.linear(x, sd[f"linears.{3 * i}.weight"], sd[f"linears.{3 * i}.bias"])
            b = f"linears.{3 * i + 1}"
            x = F.batch_norm(x, buffers[b + ".running_mean"], buffers[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"],
                             training=training, momentum=0.1, eps=1e-5)      # in place on the running statistics, as nn.BatchNorm1d
            x = relu(x)
        return F.linear(x, sd["output.weight"], sd["output.bias"])

    def _noisy(self, ctrl, forced=None):
        rows = ctrl.shape[0]
        if forced is None:
            noise = torch.randn((rows, ctrl.shape[1]))                                   # trainer.py:71
            timesteps = torch.randint(0, self.ddim.num_train_timesteps, (rows,)).long()  # :72-76
        else:
            noise, timesteps = forced
        self.draws = (noise, timesteps)
        return self.ddim.add_noise(ctrl, noise, timesteps), timesteps.float() / self.ddim.num_train_timesteps

    def step(self, ctrl, score, input_ori, input_pos, object_vertices, forced=None, replicas: int = 1):
        """replicas > 1: nn.DataParallel around the model (trainer.py:41-43) as its documentation states it - the inputs are cut with
        torch.chunk, every replica runs the module on its chunk (BatchNorm batch statistics per chunk; only the first replica's
        running-statistics updates survive, the others work on copies), the outputs are concatenated for one loss, and autograd adds
        the replicas' gradients.  No multi-GPU host exists to pin this against the real wrapper: parity unpinned for replicas > 1."""
        noisy, t = self._noisy(ctrl, forced)
        leaf = {k: self.sd[k].clone().requires_grad_(True) for k in self.names}
        if replicas == 1:
            pred = self._forward({**self.sd, **leaf}, noisy, input_ori, input_pos, t, object_vertices, True)
        else:
            outs, first = [], None
            for part in zip(*[torch.chunk(v, replicas) for v in (noisy, input_ori, input_pos, t, object_vertices)]):
                bufs = {k: v.clone() for k, v in self.sd.items() if "running_" in k}       # every replica starts from the same buffers
                outs.append(self._forward({**self.sd, **leaf}, *part, True, buffers=bufs))
                first = first or bufs
            self.sd.update(first)
            pred = torch.cat(outs)
        loss = F.mse_loss(pred, score)
        grads = torch.autograd.grad(loss, [leaf[k] for k in self.names])
        self.grads = dict(zip(self.names, grads))
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k in self.names:
            g = self.grads[k] + self.wd * self.sd[k] if self.wd else self.grads[k]
            self.m[k] = self.m[k] + (g - self.m[k]) * (1 - b1)
            self.v[k] = self.v[k] * b2 + (1 - b2) * g * g
            self.sd[k] = self.sd[k] - (self.lr / bc1) * self.m[k] / (self.v[k].sqrt() / math.sqrt(bc2) + self.eps)
        for k in self.sd:
            if k.endswith("num_batches_tracked"):
                self.sd[k] = self.sd[k] + 1
        return float(loss.detach()), pred.detach()

    def inference(self, ctrl, score, input_ori, input_pos, object_vertices, forced=None):
        # trainer.py:108-146: eval mode, same draws, no update
        with torch.no_grad():
            noisy, t = self._noisy(ctrl, forced)
            pred = self._forward(self.sd, noisy, input_ori, input_pos, t, object_vertices, False)
            return pred, float(F.mse_loss(pred, score))


# =========================================================================== (f) rank 4  training the eps-net
class EMAModel:
    """``diffusers.training_utils.EMAModel`` of diffusers 0.11.1 (requirements.txt:1; constructed at generator/diffusion.py:83-87 with
    ``power`` / ``update_after_step``, stepped from ``on_train_batch_end`` :716-720), restated from the published code over a flat
    state_dict: ``decay = 0`` while ``step <= 0`` with ``step = max(0, optimization_step - update_after_step - 1)``, else
    ``1 - (1 + step / inv_gamma) ** -power`` clamped to ``[min_value, max_value]``; every parameter ``ema = ema * decay + (1 - decay) *
    param`` (``mul_`` then ``add_(param, alpha=1 - decay)``); then ``optimization_step += 1``.  diffusers is absent from this image:
    **parity unpinned**, like the DDIM scheduler."""

    def __init__(self, sd: SD, update_after_step: int = 0, inv_gamma: float = 1.0, power: float = 2 / 3, min_value: float = 0.0,
                 max_value: float = 0.9999):
        self.averaged = {k: v.clone() for k, v in sd.items()}
        self.update_after_step, self.inv_gamma, self.power = update_after_step, inv_gamma, power
        self.min_value, self.max_value = min_value, max_value
        self.decay, self.optimization_step = 0.0, 0

    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.update_after_step - 1)
        value = 1 - (1 + step / self.inv_gamma) ** -self.power
        if step <= 0:
            return 0.0
        return max(self.min_value, min(value, self.max_value))

    def step(self, sd: SD) -> None:
        self.decay = self.get_decay(self.optimization_step)
        for k, p in sd.items():
            e = self.averaged[k]
            e.mul_(self.decay)
            e.add_(p, alpha=1 - self.decay)
        self.optimization_step += 1


class UnetTrainer:
    """Training of the eps-net as ``Diffusion`` does it under Lightning's automatic optimisation, restated functionally over a flat
    state_dict: ``get_stats`` (generator/diffusion.py:126-166: ``torch.randn`` noise then ``torch.randint`` timesteps from the CPU
    generator, ``DDIMScheduler.add_noise``, the U-Net, ``F.mse_loss(noise_pred, noise)``), ``training_step`` (:168-177), the
    optimiser of ``configure_optimizers`` (:711-714: ``torch.optim.Adam(lr)`` with torch's defaults, betas (0.9, 0.999), eps 1e-8, no
    weight decay; written out in the single-tensor form of torch 2.x) and ``on_train_batch_end`` (:716-724: ``EMAModel.step``).
    ``num_timesteps_per_batch`` is 1 (the constructor default, never overridden by generator/train.py)."""

    def __init__(self, sd: SD, num_train_timesteps: int, num_points: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 ema_power: float = 0.75, ema_update_after_step: int = 0):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = list(self.sd.keys())
        self.ddim = DDIM(num_train_timesteps)
        self.L, self.lr, self.betas, self.eps = num_points, lr, betas, eps
        self.m = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.t = 0
        self.grads: SD = {}
        self.draws = None
        self.ema = EMAModel(self.sd, update_after_step=ema_update_after_step, power=ema_power)

    def get_stats(self, x0: torch.Tensor, forced=None, sd: Optional[SD] = None):
        B = x0.shape[0]
        if forced is None:
            noise = torch.randn((B, self.L, 1))                                            # diffusion.py:134
            timesteps = torch.randint(0, self.ddim.num_train_timesteps, (B,)).long()       # :137-142
        else:
            noise, timesteps = forced
        self.draws = (noise, timesteps)
        noisy = self.ddim.add_noise(x0, noise, timesteps)                                  # :146-150
        pred = unet1d_forward(self.sd if sd is None else sd, noisy, timesteps)             # :153-157
        return F.mse_loss(pred, noise), pred                                               # :164

    def step(self, x0: torch.Tensor, forced=None):
        leaf = {k: self.sd[k].clone().requires_grad_(True) for k in self.names}
        loss, pred = self.get_stats(x0, forced, leaf)
        grads = torch.autograd.grad(loss, [leaf[k] for k in self.names])
        self.grads = dict(zip(self.names, grads))
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k in self.names:
            g = self.grads[k]
            self.m[k] = self.m[k] + (g - self.m[k]) * (1 - b1)
            self.v[k] = self.v[k] * b2 + (1 - b2) * g * g
            self.sd[k] = self.sd[k] - (self.lr / bc1) * self.m[k] / (self.v[k].sqrt() / math.sqrt(bc2) + self.eps)
        self.ema.step(self.sd)
        return float(loss.detach()), pred.detach()


# =========================================================================== (f) rank 4  Trainer.step of the 3-D dynamics model
class Trainer3D:
    """``Trainer`` (dynamics/trainer.py:16-146) for ``ProfileForward3DModel``, restated functionally over a flat state_dict: one
    ``step`` = the body of trainer.py:83-92 / :96-103 on the rows it is handed (the caller makes the ``--use_sub_batch`` slices) - the
    model in training mode: PointNet++ with BatchNorm2d batch statistics over every grouped point of the batch
    (pointnet2_utils.py:184-210), the trunk's BatchNorm1d, running statistics (momentum 0.1, unbiased variance) - nn.MSELoss,
    torch.autograd, torch.optim.Adam(lr, betas=(0.9, 0.95), weight_decay) (:46) in the single-tensor form.  ``time_encoder`` is built
    but never called by forward (profile_forward_3d.py:83): its gradient is None and Adam skips it, weight decay included.  Draws, in
    the reference's order: ``draws`` (torch.randn for channel 1's noise, :68; torch.randint timesteps, :69-73) once per batch, then per
    forward the FPS starts of sa1 and sa2 (pointnet2_utils.py:83), recorded in ``starts.log``."""

    def __init__(self, sd: SD, num_train_timesteps: int, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.95), eps: float = 1e-8):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names = [k for k, v in self.sd.items() if v.dtype.is_floating_point and "running_" not in k]
        self.ddim = DDIM(num_train_timesteps)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.names}
        self.t = 0
        self.grads: SD = {}

    def draw(self, ctrl: torch.Tensor):
        """The per-batch draws of trainer.py:68-73 for ctrl (rows, 3, L): (noise (rows, L) of channel 1, timesteps (rows,))."""
        rows, _, L = ctrl.shape
        return torch.randn((rows, 1, L))[:, 0, :], torch.randint(0, self.ddim.num_train_timesteps, (rows,)).long()

    def _noisy(self, ctrl, draws):
        noise, timesteps = draws
        full = torch.cat([torch.zeros_like(noise)[:, None], noise[:, None], torch.zeros_like(noise)[:, None]], dim=1)      # :68
        return self.ddim.add_noise(ctrl, full, timesteps), timesteps.float() / self.ddim.num_train_timesteps

    def _forward(self, sd, noisy, ori, pos, t, object_vertices, training, starts, buffers=None):
        bufs = self.sd if buffers is None else buffers
        g = _mlp2(sd, "gripper_encoder", noisy[:, 1, :], F.relu)
        pose = torch.cat([nerf_embed(ori), nerf_embed(pos)], dim=1)
        o = pointnet2_forward(sd, object_vertices, starts, prefix="object_encoder.", training=training, buffers=bufs)
        te = timestep_embedding(t, sd["gripper_encoder.2.weight"].shape[0])
        x = torch.cat([o, g, pose, te], dim=1)
        i = 0
        while f"linears.{3 * i}.weight" in sd:
            x = F.linear(x, sd[f"linears.{3 * i}.weight"], sd[f"linears.{3 * i}.bias"])
            b = f"linears.{3 * i + 1}"
            x = F.relu(F.batch_norm(x, bufs[b + ".running_mean"], bufs[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"],
                                    training=training, momentum=0.1, eps=1e-5))
            i += 1
        return F.linear(x, sd["output.weight"], sd["output.bias"])

    def step(self, ctrl, score, input_ori, input_pos, object_vertices, draws, starts: Optional[StartLog] = None, replicas: int = 1):
        """replicas > 1: nn.DataParallel around the model (trainer.py:41-43) as its documentation states it - the rows are cut with
        torch.chunk, every replica runs the module on its chunk in training mode (so every BatchNorm layer, PointNet++'s included,
        normalises with ITS chunk's statistics; only replica 0's running-statistics updates survive), the outputs are concatenated for
        one loss and autograd adds the replicas' gradients.  The replicas' FPS start draws come from the one CPU generator in thread
        order, which the reference leaves undefined; here: sa1's and sa2's draws for ALL rows first (the single-process order), each
        replica taking its chunk's.  No multi-GPU host exists to pin this against the real wrapper: parity unpinned for replicas > 1."""
        noisy, t = self._noisy(ctrl, draws)
        leaf = {k: self.sd[k].clone().requires_grad_(True) for k in self.names}
        if replicas == 1:
            pred = self._forward({**self.sd, **leaf}, noisy, input_ori, input_pos, t, object_vertices, True, starts or StartLog())
        else:
            rows, N = ctrl.shape[0], object_vertices.shape[2]
            log = starts or StartLog()
            s1, s2 = log.draw(N, rows), log.draw(512, rows)
            outs, first = [], None
            for part in zip(*[torch.chunk(v, replicas) for v in (noisy, input_ori, input_pos, t, object_vertices, s1, s2)]):
                bufs = {k: v.clone() for k, v in self.sd.items() if "running_" in k}       # every replica starts from the same buffers
                outs.append(self._forward({**self.sd, **leaf}, *part[:5], True, StartLog([part[5], part[6]]), buffers=bufs))
                first = first or bufs
            self.sd.update(first)
            pred = torch.cat(outs)
        loss = F.mse_loss(pred, score)
        grads = torch.autograd.grad(loss, [leaf[k] for k in self.names], allow_unused=True)
        self.grads = {k: g for k, g in zip(self.names, grads) if g is not None}
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k, g in self.grads.items():
            g = g + self.wd * self.sd[k] if self.wd else g
            self.m[k] = self.m[k] + (g - self.m[k]) * (1 - b1)
            self.v[k] = self.v[k] * b2 + (1 - b2) * g * g
            self.sd[k] = self.sd[k] - (self.lr / bc1) * self.m[k] / (self.v[k].sqrt() / math.sqrt(bc2) + self.eps)
        for k in self.sd:
            if k.endswith("num_batches_tracked"):
                self.sd[k] = self.sd[k] + 1
        return float(loss.detach()), pred.detach()

    def inference(self, ctrl, score, input_ori, input_pos, object_vertices, draws, starts: Optional[StartLog] = None):
        with torch.no_grad():
            noisy, t = self._noisy(ctrl, draws)
            pred = self._forward(self.sd, noisy, input_ori, input_pos, t, object_vertices, False, starts or StartLog())
            return pred, float(F.mse_loss(pred, score))
