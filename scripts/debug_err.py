import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from dgdm_amd.scheduler import DDIMScheduler
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
B,G,P,L,nv=4,10,2,14,100
objs=[synth.synth_object_2d(i,nv) for i in range(3)]
usd, dsd = util.unet_sd(11), util.dyn2d_sd(22,nv)
s=util.setup('point', usd, dsd, 15,5,L,G,P)
net, dyn = engine.Unet1d(usd), engine.Dynamics(2, dsd, L, 2*nv)
gd = engine.Guidance(dyn, B, G, P, (-1.0,1.0), 3, 15, nv, 0, 3); gd.set_objects(torch.stack(objs).to(dev))
sch = DDIMScheduler(num_train_timesteps=15); sch.set_timesteps(5)
noise=synth.synth_noise(0,B,L)
# teacher-forced along oracle trajectory of multi-object chain
x = noise.clone()
for t in s.sched.timesteps:
    ts = t*torch.ones(B,dtype=torch.int64)
    eps = orc.unet1d_forward(usd, x, ts)
    gs = [orc.cond_fn(s, x, ts, 'shift_down', o) for o in objs]
    e_h = net.forward(x.to(dev), ts.to(dev)).cpu()
    g_h = gd.grad(x.reshape(1,B,L).expand(3,-1,-1).contiguous().to(dev), int(t), [engine.make_objective('shift_down', i) for i in range(3)]).cpu()
    print("t", int(t), "eps rel", util.rel_l2(e_h, eps), "grad rel", [util.rel_l2(g_h[i].reshape(B,L,1), gs[i]) for i in range(3)], "|g|", float(gs[0].abs().max()), "|eps|", float(eps.abs().max()))
    g = sum(gs)/3
    e2 = eps - (1 - s.sched.alphas_cumprod[t]).sqrt()*g*0.001
    xn = s.sched.step(e2, t, x)
    xh = engine.ddim_guided_step(x.to(dev), e_h.to(dev), g_h.to(dev).reshape(3,-1), 3, sch.coefficients(int(t)), 0.001).cpu()
    print("   step abs err", float((xh.reshape(B,-1)-xn.reshape(B,-1)).norm(dim=1).max()))
    x = xn
