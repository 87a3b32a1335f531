"""Float64 yardstick for the 3-D guided chains, evaluated per OBJECT instead of per replicated row.  TEST INFRASTRUCTURE ONLY
(same rule as oracle/dgdm_oracle.py: only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it).

Why it exists.  ``dgdm_oracle.cond_fn`` in float64 evaluates PointNet++ on every one of the R replicated rows of a ``cond_fn`` call
(generator/diffusion.py:473-504; dynamics/models/pointnet2.py:21-32) - 25-50 minutes of CPU per full-grid chain - which is why round 3
had float64 chains for six reference chains only.  In EXACT arithmetic the embedding of a row is a function of the object and of the
row's two FPS start draws (s1, s2) alone (pointnet2_utils.py:83; DESIGN_HISTORY.md 4.3 states the decomposition):

  * sa1 (npoint = N = 512): FPS visits every point, so the 512 centre features are the per-point features F1[p] in the order
    fps1[s1]; ball query scans the ORIGINAL order (pointnet2_utils.py:95-115), so F1 does not depend on s1;
  * sa2: the pair features Y[c][k] of centre point c and neighbour point k (131 -> 128 -> 256 on [xyz_k - xyz_c | F1[k]]) depend on
    the two points only; a centre's pooled feature is the max over its first 64 in-radius neighbours IN THE ORDER fps1[s1]:
    L2[s1][c]; sa3's layer on [xyz_c | L2[s1][c]] is Z[s1][c];
  * a row: its 128 centres are FPS(128) of the cloud in the order fps1[s1] from position s2; its embedding is max_c Z[s1][c].

Every index decision (FPS picks, ball membership) is taken in float32 by dgdm_oracle's own functions, exactly as dgdm_oracle's
float64 mode takes them; every value is float64.  Float64 sums in a different order differ by 1e-16, and a ReLU pre-activation that
close to zero has probability ~1e-9 per call, so this IS the float64 evaluation of the as-written dataflow:
tests/test_oracle_fast64.py checks it against dgdm_oracle's float64 mode (embeddings 1e-13, first-step gradients and whole chains
of tests/golden/g9_f64.npz to 1e-10).  Per object the tables cost about two minutes of CPU; a chain then costs seconds.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import dgdm_oracle as orc

SD = Dict[str, torch.Tensor]


def _mlp_layer(sd: SD, p: str, i: int, rows: torch.Tensor) -> torch.Tensor:
    """mlp_convs.i (1x1 Conv2d) -> mlp_bns.i (eval) -> ReLU of pointnet2_utils.py:201-204 on rows [n][C]."""
    w = sd[f"{p}.mlp_convs.{i}.weight"]
    x = F.linear(rows, w.reshape(w.shape[0], -1), sd[f"{p}.mlp_convs.{i}.bias"])
    b = f"{p}.mlp_bns.{i}"
    x = (x - sd[b + ".running_mean"]) / torch.sqrt(sd[b + ".running_var"] + 1e-5) * sd[b + ".weight"] + sd[b + ".bias"]
    return F.relu(x)


class ObjectTables64:
    """The per-object tables above.  `sd64`: the dynamics state_dict in float64; `xyz`: the object's (512, 3) points (float32 values)."""

    def __init__(self, sd64: SD, xyz: torch.Tensor, prefix: str = "object_encoder.", s1_only: Optional[Sequence[int]] = None):
        """`s1_only`: build the per-s1 part (crowded centres) for these sa1 start draws only (tests; embed() then takes no other s1)."""
        xyz32 = xyz.float().contiguous()
        x64 = xyz32.double()
        N = xyz32.shape[0]
        self.N, self.xyz32 = N, xyz32
        p1, p2, p3 = prefix + "sa1", prefix + "sa2", prefix + "sa3"
        # T1: sa1's FPS order from every start (float32 decisions, the oracle's own function)
        self.fps1 = orc.farthest_point_sample(xyz32[None].expand(N, -1, -1).contiguous(), N, torch.arange(N))      # [s1][pos] -> point id
        # T2: sa1 features per point (ball query in the original order, centre = the point itself)
        idx1 = orc.query_ball_point(0.2, 32, xyz32[None], xyz32[None])[0]                                          # [p][32]
        rel = x64[idx1] - x64[:, None, :]
        f = _mlp_layer(sd64, p1, 1, _mlp_layer(sd64, p1, 0, rel.reshape(-1, 3)))
        self.F1 = f.reshape(N, 32, -1).max(1)[0]                                                                    # [p][128]
        # sa2 ball membership for every ordered pair, by the oracle's expanded float32 distance form
        self.M2 = ~(orc.square_distance(xyz32[None], xyz32[None])[0] > 0.4 ** 2)                                    # [c][k]
        cnt = self.M2.sum(1)
        self.crowded = cnt > 64
        # T4: pair features for the in-radius pairs, dense [c][k][256], -inf elsewhere
        ci, ki = torch.nonzero(self.M2, as_tuple=True)
        feat = torch.cat([x64[ki] - x64[ci], self.F1[ki]], dim=1)
        y = torch.empty(ci.numel(), sd64[f"{p2}.mlp_convs.1.weight"].shape[0], dtype=torch.float64)
        for a in range(0, ci.numel(), 65536):
            y[a:a + 65536] = _mlp_layer(sd64, p2, 1, _mlp_layer(sd64, p2, 0, feat[a:a + 65536]))
        W = y.shape[1]
        Y = torch.full((N, N, W), float("-inf"), dtype=torch.float64)
        Y[ci, ki] = y
        # T5/T6: centres whose ball holds <= 64 points keep all of them under any order: one row each
        l2_0 = Y.max(1)[0]                                                                                          # [c][256] (all in-radius k)
        z = lambda l2, cid: _mlp_layer(sd64, p3, 0, torch.cat([x64[cid], l2], dim=1))                              # noqa: E731
        self.Z0 = z(l2_0, torch.arange(N))                                                                          # valid where not crowded
        # crowded centres: the first 64 in-radius points in the order fps1[s1], per s1
        self.cr_ids = torch.nonzero(self.crowded).reshape(-1)
        self.cr_slot = torch.full((N,), -1, dtype=torch.long)
        self.cr_slot[self.cr_ids] = torch.arange(self.cr_ids.numel())
        ncr = self.cr_ids.numel()
        self.Zc = torch.full((N, ncr, self.Z0.shape[1]), float("nan"), dtype=torch.float64)                        # [s1][slot][256]
        if ncr:
            Ycr, Mcr = Y[self.cr_ids], self.M2[self.cr_ids]                                                         # [ncr][k][256], [ncr][k]
            for s1 in (range(N) if s1_only is None else s1_only):
                order = self.fps1[s1]
                mo = Mcr[:, order]
                sel_pos = mo & (mo.cumsum(1) <= 64)                                                                 # [ncr][pos]
                sel_k = torch.zeros_like(sel_pos)
                sel_k[:, order] = sel_pos
                l2 = Ycr.masked_fill(~sel_k[:, :, None], float("-inf")).max(1)[0]
                self.Zc[s1] = z(l2, self.cr_ids)

    def embed(self, s1: torch.Tensor, s2: torch.Tensor) -> torch.Tensor:
        """Embeddings [rows][256] (float64) of the rows whose sa1 / sa2 FPS start draws are s1 / s2."""
        perm = self.fps1[s1]                                                                                        # [rows][512] point ids
        cloud = self.xyz32[perm]                                                                                    # the cloud sa2 sees, float32
        fps2 = orc.farthest_point_sample(cloud, 128, s2)                                                            # positions in the permuted cloud
        cid = torch.gather(perm, 1, fps2)                                                                           # [rows][128] centre point ids
        zr = self.Z0[cid]                                                                                           # [rows][128][256]
        slot = self.cr_slot[cid]
        rr, jj = torch.nonzero(slot >= 0, as_tuple=True)
        if rr.numel():
            zr[rr, jj] = self.Zc[s1[rr], slot[rr, jj]]
        return zr.max(1)[0]


def dyn3d_logits(sd: SD, x_ctrl, x_ori, x_pos, timesteps, emb: torch.Tensor) -> torch.Tensor:
    """``ProfileForward3DModel.forward`` (dynamics/profile_forward_3d.py:67-86) with the object embedding given: the lines of
    dgdm_oracle.dyn3d_forward after pointnet2_forward."""
    g = orc._mlp2(sd, "gripper_encoder", x_ctrl[:, 1, :], F.relu)
    pose = torch.cat([orc.nerf_embed(x_ori), orc.nerf_embed(x_pos)], dim=1)
    W = sd["gripper_encoder.2.weight"].shape[0]
    te = orc.timestep_embedding(timesteps, W)
    return orc._trunk(sd, torch.cat([emb, g, pose, te], dim=1).to(emb.dtype), n_embed=emb.shape[1])


def cond_fn(s: orc.Setup, tab: ObjectTables64, x: torch.Tensor, t: torch.Tensor, opt_obj: str, centers, calls: Sequence[torch.Tensor],
            tiles: Optional[list] = None, risk: Optional[list] = None, risk_level: float = 2.0 ** -20) -> torch.Tensor:
    """``Diffusion.cond_fn`` for 'point_3d' (generator/diffusion.py:473-504) in float64 with the recorded FPS draws of this call
    (`calls` = [sub-batch 0's sa1 draws, its sa2 draws, sub-batch 1's ...], the order dgdm_oracle.StartLog replays them in).
    `tiles`: when a list, receives the [B][tiles][W1] sums of d objective / d z1 over 32 consecutive pose cells per finger - what the
    HIP trunk leaves behind per 32-row tile (make_golden.g9_tiles).
    `risk` (with `tiles`): when a list, receives (number of rows holding a ReLU whose float64 input lies within `risk_level` - relative to
    the size of the sum that produces it, dgdm_oracle._relu_margin - of zero, the sum over those rows of the norm of the row's whole
    contribution to its finger's gradient): what float32 sign flips can move, from float64 evidence alone (make_golden.g9_ties64)."""
    T = s.sched.num_train_timesteps
    with torch.enable_grad():
        x = x.detach().double().requires_grad_(True)
        B = x.shape[0]
        cells = s.grid_size * s.num_pos ** 2
        ori, pos = orc._pose_grid(s, B, (-1.0, 1.0))
        tt = t.repeat(cells).float() / T
        pts = orc._pts3d(s, x).repeat(cells, 1, 1)
        kw = dict(centers=centers, grid_size=s.grid_size, num_pos=s.num_pos)
        grad, k, zrows, margins = 0.0, 0, [], []
        for i in range(0, B * cells, s.sub_batch_size):
            j = min(i + s.sub_batch_size, B * cells)
            s1, s2 = calls[k], calls[k + 1]
            k += 2
            assert s1.shape == (j - i,) and s2.shape == (j - i,)
            emb = tab.embed(s1, s2)
            if tiles is not None:
                orc.TRUNK_CAPTURE = []
                if risk is not None:
                    orc.TRUNK_MARGIN = []
            logits = dyn3d_logits(s.dyn, pts[i:j], ori[i:j], pos[i:j], tt[i:j], emb)
            val = orc.deltas_to_objective(logits, opt_obj, **kw).sum()
            if tiles is not None:
                z1 = orc.TRUNK_CAPTURE[0]
                orc.TRUNK_CAPTURE = None
                if risk is not None:
                    margins.append(orc.TRUNK_MARGIN[0])
                    orc.TRUNK_MARGIN = None
                gx, gz = torch.autograd.grad(val, [x, z1])
                zrows.append(gz.detach())
            else:
                gx = torch.autograd.grad(val, x)[0]
            grad = grad + gx
        if tiles is not None:
            zg = torch.cat(zrows)                                    # [R][W1], reference row r = cell * B + b
            nt = (cells + 31) // 32
            acc = torch.zeros(B, nt, zg.shape[1], dtype=torch.float64)
            for b in range(B):
                zb = zg[torch.arange(cells) * B + b]
                for ti in range(nt):
                    acc[b, ti] = zb[32 * ti:32 * ti + 32].sum(0)
            tiles.append(acc)
            if risk is not None:
                mg = torch.cat(margins)
                rows = torch.nonzero(mg < risk_level).reshape(-1)
                total = 0.0
                jac = {}
                for r in rows.tolist():
                    b = r % B
                    if b not in jac:
                        jac[b] = gripper_jacobian(s.dyn, x.detach()[b, :, 0])
                    total += float((zg[r] @ jac[b]).norm())
                risk.append((int(rows.numel()), total))
        return grad


def gripper_jacobian(sd64: SD, xb: torch.Tensor) -> torch.Tensor:
    """d z1 / d x for one finger (x: the control values that reach the model, channel 1 of x_ctrl): z1 = the first trunk layer's
    pre-activation (after its eval-mode BatchNorm), whose x-dependent part is W1[:, 256:512] . gripper_encoder(x)
    (dynamics/profile_forward_3d.py:77,84).  [W1][L], float64.  A tile's share of the finger's gradient is (sum over the tile's rows of
    d objective / d z1) @ this."""
    w0, b0, w2 = sd64["gripper_encoder.0.weight"], sd64["gripper_encoder.0.bias"], sd64["gripper_encoder.2.weight"]
    act = ((w0 @ xb + b0) > 0).double()
    sc = sd64["linears.1.weight"] / torch.sqrt(sd64["linears.1.running_var"] + 1e-5)
    W = w2.shape[0]
    return (sc[:, None] * sd64["linears.0.weight"][:, W:2 * W]) @ (w2 * act[None, :]) @ w0


def _f64(sd: SD) -> SD:
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def guided_chain(unet_sd: SD, dyn_sd: SD, sched: orc.DDIM, L: int, G: int, P: int, sub_bs: int, noise: torch.Tensor,
                 tabs: Sequence[ObjectTables64], opt_obj: str, centers, calls: List[torch.Tensor], multi: bool = False,
                 trace: Optional[list] = None) -> torch.Tensor:
    """The whole guided chain in float64 on recorded draws: generator/diffusion.py:570-576 (one object) or :637-647 (multi-object: the
    mean of the objects' gradients, object after object within a step).  `calls`: the chain's recorded FPS draws in the order the
    reference consumed them, WITHOUT the centre sweep's.  `trace`: receives (x, eps, [grad per object]) per step."""
    u64, d64 = _f64(unet_sd), _f64(dyn_sd)
    s = orc.Setup('point_3d', u64, d64, sched, L, G, P, sub_bs)
    B = noise.shape[0]
    n_sub = 2 * ((B * G * P * P + sub_bs - 1) // sub_bs)
    scale = orc.classifier_scale('point_3d', opt_obj, multi)
    x = noise.double().clone()
    k = 0
    for t in sched.timesteps:
        ts = t * torch.ones(B, dtype=torch.int64)
        with torch.no_grad():
            eps = orc.unet1d_forward(u64, x, ts)
        grads = []
        for tab in tabs:
            grads.append(cond_fn(s, tab, x, ts, opt_obj, centers, calls[k:k + n_sub]))
            k += n_sub
        gr = grads[0] if not multi else torch.stack(grads).mean(0)
        if trace is not None:
            trace.append((x.clone(), eps.clone(), [g.clone() for g in grads]))
        eps = eps - (1 - s.sched.alphas_cumprod[t]).sqrt() * gr * scale
        x = s.sched.step(eps, t, x)
    assert k == len(calls), (k, len(calls))
    return x
