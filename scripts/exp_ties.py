"""Where do the ReLU ties of the 3-D trunk come from?  CPU experiment on the first cond_fn call of tests/golden/g9_3d_<part>.npz.

Counts, per trunk layer, the units whose pre-activation has a different SIGN than in an all-float64 evaluation, for
  ref      torch float32 trunk on the torch float32 PointNet++ embedding            (= the reference's arithmetic)
  seq      the HIP trunk's arithmetic emulated (BatchNorm folded in float64 and rounded, one k-ordered float32 fma chain per
           output, layer 1 as chain(W1o xobj) + (Atab + Ptab)) on the float32 embedding
  seq+e    the same on the embedding perturbed to the table pipeline's measured error (1.7e-6 relative, DESIGN_HISTORY.md 7)
  seq+x    the same on the float64 embedding rounded once to float32 (what a float64 table build would deliver)
  blkN     as seq+x with every chain split into N interleaved partial chains summed at the end
and the disagreements of each variant WITH THE REFERENCE (what the parity tests see).

    python scripts/exp_ties.py [part] [cache.npz]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgdm_amd import synth                     # noqa: E402
from oracle import dgdm_oracle as orc          # noqa: E402
from tests import util as tu                   # noqa: E402

part = sys.argv[1] if len(sys.argv) > 1 else "rotate"
cache = sys.argv[2] if len(sys.argv) > 2 else f"/tmp/exp_ties_{part}.npz"
g = np.load(os.path.join(tu.GOLDEN, f"g9_3d_{part}.npz"))
B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
gain = float(g["gain"])
sd32 = synth.scale_output(tu.dyn3d_sd(g["dyn3d_seed"]), gain)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
sch = orc.DDIM(T)
sch.set_timesteps(S)
cells = G * P * P
R = B * cells
calls = tu.unpack_starts(g["starts"].astype(np.int64), g["start_lens"])
if str(g["opt_obj"]) == 'convergence':
    calls = calls[2:]
s_ = orc.Setup('point_3d', None, sd32, sch, L, G, P, 512)
x = torch.from_numpy(g["trace_x"][0])
ori, pos = orc._pose_grid(s_, B, (-1.0, 1.0))
tt = torch.full((B,), int(sch.timesteps[0]), dtype=torch.int64).repeat(cells).float() / T
pts = orc._pts3d(s_, x).repeat(cells, 1, 1)
obj = torch.from_numpy(g["objs"][0])

if os.path.exists(cache):
    c = np.load(cache)
    o32, o64 = torch.from_numpy(c["o32"]), torch.from_numpy(c["o64"])
else:
    res = {}
    for dt, sd, ov in (("32", sd32, obj), ("64", sd64, obj.double())):
        log = orc.StartLog(list(calls[:2 * ((R + 511) // 512)]))
        rows = []
        t0 = time.time()
        with torch.no_grad():
            for i in range(0, R, 512):
                j = min(i + 512, R)
                rows.append(orc.pointnet2_forward(sd, ov.t().unsqueeze(0).expand(j - i, -1, -1), log, prefix="object_encoder."))
        res[dt] = torch.cat(rows)
        print("pointnet float" + dt, f"{time.time() - t0:.0f}s", flush=True)
    o32, o64 = res["32"], res["64"]
    np.savez(cache, o32=o32.numpy(), o64=o64.numpy())
print("embedding: torch float32 vs float64 rel", tu.rel_l2(o32, o64))


def rest(sd, dt):
    gr = orc._mlp2(sd, "gripper_encoder", pts[:, 1, :].to(dt), F.relu)
    pose = torch.cat([orc.nerf_embed(ori), orc.nerf_embed(pos)], dim=1).to(dt)
    te = orc.timestep_embedding(tt, 256).to(dt)
    return gr, pose, te


def trunk_torch(sd, xin):
    pre = []
    i = 0
    h = xin
    while f"linears.{3 * i}.weight" in sd:
        h = F.linear(h, sd[f"linears.{3 * i}.weight"], sd[f"linears.{3 * i}.bias"])
        b = f"linears.{3 * i + 1}"
        h = F.batch_norm(h, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"], training=False, eps=1e-5)
        pre.append(h)
        h = F.relu(h)
        i += 1
    return pre


def chain(acc, w, h, nblk=1):
    """acc[rows, out] + sum_k w[out, k] h[rows, k] as nblk interleaved k-ordered float32 fma chains (chain j takes k = j mod nblk),
    the partial chains added in float64 and rounded once."""
    K = w.shape[1]
    wd, hd = w.double(), h.double()
    if nblk == 1:
        for k in range(K):
            acc = (acc.double() + hd[:, k:k + 1] * wd[None, :, k]).float()
        return acc
    parts = []
    for j in range(nblk):
        a = torch.zeros_like(acc)
        for k in range(j, K, nblk):
            a = (a.double() + hd[:, k:k + 1] * wd[None, :, k]).float()
        parts.append(a.double())
    return (acc.double() + sum(parts)).float()


def trunk_seq(sd, o, gr, pose, te, nblk=1):
    pre = []
    w, b = orc._folded(sd, 0)
    z_obj = chain(torch.zeros(o.shape[0], w.shape[0]), w[:, :256], o, nblk)
    atab = chain(b[None].expand(o.shape[0], -1).contiguous(), w[:, 256:512], gr)
    atab = chain(atab, w[:, 539:], te)
    ptab = chain(torch.zeros_like(atab), w[:, 512:539], pose)
    z = z_obj + (atab + ptab)
    pre.append(z)
    h = F.relu(z)
    i = 1
    while f"linears.{3 * i}.weight" in sd:
        w, b = orc._folded(sd, i)
        z = chain(b[None].expand(h.shape[0], -1).contiguous(), w, h, nblk)
        pre.append(z)
        h = F.relu(z)
        i += 1
    return pre


with torch.no_grad():
    g32, p32, t32 = rest(sd32, torch.float32)
    g64, p64, t64 = rest(sd64, torch.float64)
    exact = trunk_torch(sd64, torch.cat([o64, g64, p64, t64], dim=1))
    variants = {"ref": trunk_torch(sd32, torch.cat([o32, g32, p32, t32], dim=1))}
    variants["ref+x"] = trunk_torch(sd32, torch.cat([o64.float(), g32, p32, t32], dim=1))
    variants["seq"] = trunk_seq(sd32, o32, g32, p32, t32)
    torch.manual_seed(1)
    oe = (o64 * (1 + 1.7e-6 * torch.randn_like(o64))).float()
    variants["seq+e"] = trunk_seq(sd32, oe, g32, p32, t32)
    variants["seq+x"] = trunk_seq(sd32, o64.float(), g32, p32, t32)
    for nb in (4, 16):
        variants[f"blk{nb}+x"] = trunk_seq(sd32, o64.float(), g32, p32, t32, nb)

print(f"{R} rows; units per layer: {[int(p.shape[1]) for p in exact]}")
for name, pre in variants.items():
    vs64 = [int(((a > 0) != (e > 0)).sum()) for a, e in zip(pre, exact)]
    vsref = [int(((a > 0) != (r > 0)).sum()) for a, r in zip(pre, variants["ref"])]
    err = [float((a.double() - e).norm() / e.norm()) for a, e in zip(pre, exact)]
    print(f"{name:8s} sign flips vs float64 {vs64} = {sum(vs64):3d} | vs ref {vsref} = {sum(vsref):3d} | pre-activation rel err {['%.1e' % v for v in err]}")


def trunk_seq_exact_tables(sd, o, nblk=1):
    """seq+x with the row-invariant first-layer terms (gripper encoder, A and P tables) computed in float64 and rounded once."""
    pre = []
    w, b = orc._folded(sd, 0)
    z_obj = chain(torch.zeros(o.shape[0], w.shape[0]), w[:, :256], o, nblk)
    w64, b64 = [t_.double() for t_ in orc._folded(sd64, 0)]      # folded from the same float32 parameters, kept in float64
    sc = sd64["linears.1.weight"] / torch.sqrt(sd64["linears.1.running_var"] + 1e-5)
    w64 = sc[:, None] * sd64["linears.0.weight"]
    b64 = sc * (sd64["linears.0.bias"] - sd64["linears.1.running_mean"]) + sd64["linears.1.bias"]
    tab = (F.linear(torch.cat([g64, p64, t64], dim=1), w64[:, 256:], b64)).float()
    z = z_obj + tab
    pre.append(z)
    h = F.relu(z)
    i = 1
    while f"linears.{3 * i}.weight" in sd:
        w, b = orc._folded(sd, i)
        z = chain(b[None].expand(h.shape[0], -1).contiguous(), w, h, nblk)
        pre.append(z)
        h = F.relu(z)
        i += 1
    return pre


with torch.no_grad():
    more = {"seq+x+t": trunk_seq_exact_tables(sd32, o64.float()), "blk16+x+t": trunk_seq_exact_tables(sd32, o64.float(), 16)}
for name, pre in more.items():
    vs64 = [int(((a > 0) != (e > 0)).sum()) for a, e in zip(pre, exact)]
    vsref = [int(((a > 0) != (r > 0)).sum()) for a, r in zip(pre, variants["ref"])]
    err = [float((a.double() - e).norm() / e.norm()) for a, e in zip(pre, exact)]
    print(f"{name:8s} sign flips vs float64 {vs64} = {sum(vs64):3d} | vs ref {vsref} = {sum(vsref):3d} | pre-activation rel err {['%.1e' % v for v in err]}")
