import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0); dev = torch.device("cuda:0")
torch.set_num_threads(32)
B, G, P, L, T, S, sub = 2, 16, 3, 42, 15, 5, 64
dsd = util.dyn3d_sd(33)
obj = synth.synth_object_3d(5)
dyn = engine.Dynamics(3, dsd, L)
s32 = util.setup('point_3d', None, dsd, T, S, L, G, P, sub)
x = synth.synth_noise(0, B, L)
torch.manual_seed(1); st = sampler.StartStream(512, sub).call(B * G * P * P)
R = B * G * P * P
s1 = np.empty(R, np.int64); s2 = np.empty(R, np.int64)
for r0 in range(0, R, sub):
    n = min(sub, R - r0); s1[r0:r0+n] = st[2*r0:2*r0+n]; s2[r0:r0+n] = st[2*r0+n:2*r0+2*n]
ori, pos = orc._pose_grid(s32, B, (-1.0, 1.0))
pts = orc._pts3d(s32, x).repeat(G * P * P, 1, 1)
tt = torch.full((R,), 12.0 / T)
clouds = obj.t().unsqueeze(0).expand(R, -1, -1).contiguous()
ref = orc.dyn3d_forward(dsd, pts, ori, pos, tt, clouds, orc.StartLog([torch.from_numpy(s1), torch.from_numpy(s2)]))
got = dyn.forward3d(pts.to(dev), ori.to(dev), pos.to(dev), tt.to(dev), clouds.to(dev), torch.from_numpy(s1), torch.from_numpy(s2)).cpu()
err = (got - ref).abs().max(dim=1).values / ref.abs().max()
bad = torch.nonzero(err > 1e-5).reshape(-1)
print("rows", R, "max rel err", float(err.max()), "bad rows", bad.tolist()[:20], "their (s1,s2)", [(int(s1[i]), int(s2[i])) for i in bad.tolist()[:10]])
# embedding only
emb_ref = orc.pointnet2_forward(dsd, clouds, orc.StartLog([torch.from_numpy(s1), torch.from_numpy(s2)]), prefix="object_encoder.")
emb = dyn.pointnet2(clouds.to(dev), torch.from_numpy(s1), torch.from_numpy(s2)).cpu()
e2 = (emb - emb_ref).abs().max(dim=1).values
print("embedding: max abs err", float(e2.max()), "rows > 1e-5:", torch.nonzero(e2 > 1e-5).reshape(-1).tolist()[:20])
