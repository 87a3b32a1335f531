"""Free-running 3-D guided chain at a moderate grid (R = B*G*P^2 rows) against the CPU oracle: how far apart do the end points land?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from dgdm_amd.scheduler import DDIMScheduler
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0); dev = torch.device("cuda:0")
torch.set_num_threads(32)
B, G, P, L, T, S, sub = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 3, 42, 15, 5, 64
usd, dsd = util.unet_sd(11), util.dyn3d_sd(33)
obj = synth.synth_object_3d(5)
net, dyn = engine.Unet1d(usd), engine.Dynamics(3, dsd, L)
gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 1, T, 512, sub, 1); gd.set_objects(obj[None].to(dev))
sch = DDIMScheduler(num_train_timesteps=T); sch.set_timesteps(S)
s = util.setup('point_3d', usd, dsd, T, S, L, G, P, sub)
noise = synth.synth_noise(0, B, L)
for o in ('shift_up', 'rotate'):
    torch.manual_seed(1)
    tr = []
    out = sampler.guided_chains(net, gd, sch, 'point_3d', noise.to(dev), [(0, o)], trace=tr).cpu()[0]
    torch.manual_seed(1)
    t0 = time.time(); tro = []
    ref = orc.guided_sample(s, noise, obj, o, trace=tro)
    print(o, "R =", gd.rows, "oracle %.0fs" % (time.time() - t0), "finger L2:", (out - ref).reshape(B, -1).norm(dim=1).tolist(),
          "| step-0 grad rel err", util.rel_l2(tr[0][1][0].cpu().reshape(B, L, 1), tro[0][1]), "|g|max", float(tro[0][1].abs().max()))
