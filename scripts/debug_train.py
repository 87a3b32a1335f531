"""Per-tensor comparison of the HIP trainer's first-step gradients / updated parameters with the oracle (debugging aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgdm_oracle as orc
from tests import util
from tests.test_gpu_train import _args
from dgdm_amd import _lib
from dynamics.trainer import Trainer
_lib.device_init(0)
sd = util.dyn2d_sd(41, 100)
data = util.train2d_data(int(sys.argv[1]) if len(sys.argv) > 1 else 5, 3, int(sys.argv[2]) if len(sys.argv) > 2 else 128)
o = orc.Trainer2D(sd, 15, 1e-4)
torch.manual_seed(1234); lo, po = o.step(*data)
t = Trainer(_args(0.0)); t.create_model(state_dict=sd)
torch.manual_seed(1234); lh, ph = t.step(*data)
print("loss", lo, lh, "pred", util.rel_l2(ph.cpu(), po))
gh, sh = t.gradients(), t.state_dict()
for k, ref in o.grads.items():
    rms = float(ref.double().pow(2).mean().sqrt())
    e = float((gh[k].double() - ref.double()).abs().max())
    ep = float((sh[k].double() - o.sd[k].double()).abs().max())
    d = (gh[k].double() - ref.double()).abs().flatten() / max(rms, 1e-30)
    print(f"{k:28s} grad rms {rms:.3e} maxerr/rms {e / max(rms, 1e-30):.2e} relL2 {util.rel_l2(gh[k], ref):.2e} n(err>1e-4 rms) {int((d > 1e-4).sum())}/{d.numel()}  param maxdiff {ep:.2e}")
for k in o.sd:
    if "running" in k:
        print(k, float((sh[k] - o.sd[k]).abs().max()))
