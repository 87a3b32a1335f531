import sys; sys.path.insert(0, '.')
import torch, numpy as np, itertools
from oracle import dgdm_oracle as orc
from tests import util
from dgdm_amd.dynamics.models import pointnet2_utils as pu
ctrl, score, ori, pos, obj = util.train3d_data(7)
xyz = obj.permute(0, 2, 1).contiguous()
R = xyz.shape[0]
torch.manual_seed(5); s1 = torch.randint(0, 512, (R,))
nx = orc._gather(xyz, orc.farthest_point_sample(xyz, 512, s1))
hip = pu.square_distance(nx.cuda(), xyz.cuda()).cpu().numpy()
ref = orc.square_distance(nx, xyz).numpy()
print("hip vs oracle mismatches", int((hip != ref).sum()))
a = nx.numpy().astype(np.float64)[:, :, None, :]; b = xyz.numpy().astype(np.float64)[:, None, :, :]
f32 = lambda x: x.astype(np.float32).astype(np.float64)
fma = lambda x, y, c: f32(x * y + c)
dots = {"fma012": fma(a[..., 2], b[..., 2], fma(a[..., 1], b[..., 1], f32(a[..., 0] * b[..., 0]))),
        "unfused": f32(f32(f32(a[..., 0] * b[..., 0]) + f32(a[..., 1] * b[..., 1])) + f32(a[..., 2] * b[..., 2]))}
def sq(v, kind):
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    if kind == "plain": return f32(f32(f32(x * x) + f32(y * y)) + f32(z * z))
    if kind == "fma_in": return f32(fma(x, x, f32(y * y)) + f32(z * z))
    if kind == "fma_all": return fma(z, z, fma(y, y, f32(x * x)))
    if kind == "fma_out": return fma(z, z, f32(f32(x * x) + f32(y * y)))
for dk, dv in dots.items():
    for sk in ("plain", "fma_in", "fma_all", "fma_out"):
        cn, pn = sq(a, sk), sq(b, sk)
        for comb in ("sep", "fma1"):
            if comb == "sep": d = f32(f32(f32(-2.0 * dv) + cn) + pn)
            else: d = f32(f32(-2.0 * dv + cn) + pn)
            print(dk, sk, comb, "vs hip:", int((d.astype(np.float32) != hip).sum()), " vs oracle:", int((d.astype(np.float32) != ref).sum()))
