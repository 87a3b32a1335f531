import sys; sys.path.insert(0, '.')
import numpy as np, torch
from tests import util
from oracle import dgdm_oracle as orc
from dgdm_amd.dynamics.models import pointnet2_utils as pu
ctrl, score, ori, pos, obj = util.train3d_data(7)
xyz = obj.permute(0, 2, 1).contiguous()
R = xyz.shape[0]
torch.manual_seed(5)
s1 = torch.randint(0, 512, (R,))
fps = orc.farthest_point_sample(xyz, 512, s1)
nx = orc._gather(xyz, fps)
ref = orc.query_ball_point(0.2, 32, xyz, nx)
hip = pu.query_ball_point(0.2, 32, xyz.cuda(), nx.cuda()).cpu()
bad = (ref != hip).nonzero()
print("mismatches", bad.shape[0])
d = orc.square_distance(nx, xyz)
for b, s, n in bad[:6].tolist():
    print(b, s, n, "ref", ref[b, s].tolist()[:12], "hip", hip[b, s].tolist()[:12])
    k1, k2 = int(ref[b, s, n]), int(hip[b, s, n])
    print("   dist ref pick %.9g hip pick %.9g  r2 %.9g" % (float(d[b, s, k1]), float(d[b, s, k2]), float(np.float32(0.2 ** 2))))
hs = pu.square_distance(nx.cuda(), xyz.cuda()).cpu()
print("square_distance equal:", bool((hs == d).all()), int((hs != d).sum()), "decisions differ:", int(((hs > 0.2 ** 2) != (d > 0.2 ** 2)).sum()))
