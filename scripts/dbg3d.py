import sys; sys.path.insert(0, '.')
import argparse, numpy as np, torch, ctypes as C
from tests import util
from tests.test_gpu_train3d import _args
from dgdm_amd.dynamics.trainer import Trainer
from dgdm_amd._lib import lib, check, dptr, stream_ptr
from oracle import dgdm_oracle as orc
import torch.nn.functional as F
sd = util.dyn3d_sd(43)
data = util.train3d_data(7)
ctrl, score, ori, pos, obj = data
R = ctrl.shape[0]
t = Trainer(_args(False, 0.0)); t.create_model(sd)
torch.manual_seed(5)
lh, ph = t.step(*data)
def rd(which, shape, dtype=torch.float32):
    out = torch.empty(shape, dtype=dtype, device='cuda')
    check(lib().dgdm_trainer3d_debug_read(t._h, which, dptr(out), out.numel(), stream_ptr())); torch.cuda.synchronize()
    return out.cpu()
# oracle pieces
o = orc.Trainer3D(sd, 15, 1e-4)
torch.manual_seed(5)
draws = o.draw(ctrl); log = orc.StartLog()
noisy, tt = o._noisy(ctrl, draws)
s = {k: v.clone() for k, v in sd.items()}
xyz = obj
l1x, l1p = orc.set_abstraction(s, "object_encoder.sa1", xyz, None, 512, 0.2, 32, log, training=True, buffers={k: v.clone() for k, v in sd.items()})
l2x, l2p = orc.set_abstraction(s, "object_encoder.sa2", l1x, l1p, 128, 0.4, 64, log, training=True, buffers={k: v.clone() for k, v in sd.items()})
_, l3p = orc.set_abstraction(s, "object_encoder.sa3", l2x, l2p, None, None, None, None, training=True, buffers={k: v.clone() for k, v in sd.items()})
print("starts", [l[:4].tolist() for l in log.log])
fps1 = rd(6, (R, 512), torch.int32)
ofps = orc.farthest_point_sample(xyz.permute(0, 2, 1).contiguous(), 512, log.log[0])
print("fps1 equal", bool((fps1.long() == ofps).all()))
nx1 = rd(5, (R, 512, 3)); print("nx1", float((nx1 - l1x.permute(0, 2, 1)).abs().max()))
hl1p = rd(1, (R, 512, 128)); print("l1p", util.rel_l2(hl1p, l1p.permute(0, 2, 1)))
hl2p = rd(2, (R, 128, 256)); print("l2p", util.rel_l2(hl2p, l2p.permute(0, 2, 1)))
X0 = rd(0, (R, 800))
print("emb", util.rel_l2(X0[:, :256], l3p.reshape(R, -1)))
g = orc._mlp2(s, "gripper_encoder", noisy[:, 1, :], F.relu); print("gripper", util.rel_l2(X0[:, 256:512], g))
pose = torch.cat([orc.nerf_embed(ori), orc.nerf_embed(pos)], dim=1); print("pose", float((X0[:, 512:539] - pose).abs().max()))
te = orc.timestep_embedding(tt, 256); print("time", float((X0[:, 539:795] - te).abs().max()), "pad", float(X0[:, 795:].abs().max()))
torch.manual_seed(5)
lo, po = o.step(*data, o.draw(ctrl), orc.StartLog())
print("loss", lh, lo, "pred", util.rel_l2(ph.cpu(), po))
# sa1 internals
xyzp = xyz.permute(0, 2, 1).contiguous()
idx = orc.query_ball_point(0.2, 32, xyzp, l1x.permute(0, 2, 1).contiguous())
hidx = rd(7, (R, 512, 32), torch.int32)
print("idx1 equal", bool((hidx.long() == idx).all()), int((hidx.long() != idx).sum()))
feat = orc._gather(xyzp, idx) - l1x.permute(0, 2, 1).reshape(R, 512, 1, 3)
hf = rd(3, (R, 512, 32, 4))
print("feat1", float((hf[..., :3] - feat).abs().max()), float(hf[..., 3].abs().max()))
w = sd["object_encoder.sa1.mlp_convs.0.weight"].reshape(64, 3); b = sd["object_encoder.sa1.mlp_convs.0.bias"]
y = feat.reshape(-1, 3) @ w.t() + b
hy = rd(4, (R * 512 * 32, 64))
print("y11", util.rel_l2(hy, y))
