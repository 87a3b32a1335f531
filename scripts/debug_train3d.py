#!/usr/bin/env python3
"""MI355X: per-tensor distance of the 3-D trainer's first-step gradients from the float64 oracle (and of torch's float32 autograd from
it) at a given number of rows and of distinct clouds.  usage: python scripts/debug_train3d.py [rows] [clouds] [seed] [first cloud id] [data seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import util
from tests.test_gpu_train3d import _args
from tests import train3d_common as t3
from dgdm_amd.dynamics.trainer import Trainer
from dgdm_amd import synth
from oracle import dgdm_oracle as orc

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
clouds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 57
base = int(sys.argv[4]) if len(sys.argv) > 4 else 80
dseed = int(sys.argv[5]) if len(sys.argv) > 5 else rows
sd = util.dyn3d_sd(seed)
rs = np.random.RandomState(dseed)
ctrl = torch.from_numpy(rs.uniform(-1, 1, (rows, 3, 42)).astype(np.float32))
obj = torch.stack([synth.synth_object_3d(base + i % clouds) for i in range(rows)]).permute(0, 2, 1).contiguous()
ori = torch.from_numpy(rs.uniform(-1, 1, (rows, 1)).astype(np.float32))
pos = torch.from_numpy(rs.uniform(-1, 1, (rows, 2)).astype(np.float32))
score = torch.from_numpy(rs.normal(0, 1, (rows, 3)).astype(np.float32))
o = orc.Trainer3D(sd, 15, 1e-4, 0.0)
torch.manual_seed(rows)
draws, log = o.draw(ctrl), orc.StartLog()
lo, po = o.step(ctrl, score, ori, pos, obj, draws, log)
o64 = orc.Trainer3D({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, 15, 1e-4, 0.0)
l64, p64 = o64.step(ctrl.double(), score.double(), ori.double(), pos.double(), obj.double(), (draws[0].double(), draws[1]), orc.StartLog(list(log.log)))
t = Trainer(_args(False, 0.0))
t.create_model(sd)
torch.manual_seed(rows)
lh, ph = t.step(ctrl, score, ori, pos, obj)
print(f"rows {rows} clouds {clouds}: loss HIP {lh:.7f} f32 {lo:.7f} f64 {l64:.7f}; pred HIP-f64 {util.rel_l2(ph.cpu().double(), p64):.1e} f32-f64 {util.rel_l2(po.double(), p64):.1e}")
gh = t.gradients()
for k in o.grads:
    e_h, e_o = util.rel_l2(gh[k].double(), o64.grads[k]), util.rel_l2(o.grads[k].double(), o64.grads[k])
    e_ho = util.rel_l2(gh[k].double(), o.grads[k].double())
    flag = " (BN-fed bias)" if k in t3.BN_FED_BIAS else ""
    if (e_h > 1e-4 or e_o > 1e-4) and k not in t3.BN_FED_BIAS:
        print(f"  {k:50s} HIP-f64 {e_h:.1e}  f32-f64 {e_o:.1e}  HIP-f32 {e_ho:.1e}  rms {float(o64.grads[k].pow(2).mean().sqrt()):.2e}{flag}")
