import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, _lib
from dgdm_amd.scheduler import DDIMScheduler
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0); dev = torch.device("cuda:0")
g=util.load("g6_chains.npz")
xs,es,gs=[torch.from_numpy(g["trace3d_rotate"+k]) for k in ("_x","_eps","_grad")]
so=orc.DDIM(15); so.set_timesteps(5)
s=DDIMScheduler(num_train_timesteps=15); s.set_timesteps(5)
for si,t in enumerate(so.timesteps):
    t=int(t)
    sa,sb,sap,sbp = s.coefficients(t)
    x,e,gr = xs[si],es[si],gs[si]
    e2 = e - torch.tensor(sb)*gr*0.5
    e2_ref = e - (1 - so.alphas_cumprod[t]).sqrt()*gr*0.5
    x0 = ((x - torch.tensor(sb)*e2)/torch.tensor(sa)).clamp(-1,1)
    nxt = torch.tensor(sap)*x0 + torch.tensor(sbp)*e2
    want = so.step(e2_ref, t, x)
    h = engine.ddim_guided_step(x.to(dev), e.to(dev), gr.to(dev).reshape(1,-1), 1, (sa,sb,sap,sbp), 0.5).cpu()
    # hip pieces: only combine (scale path) by passing sap=0,sbp=1: out = e2
    e2_h = engine.ddim_guided_step(x.to(dev), e.to(dev), gr.to(dev).reshape(1,-1), 1, (1.0,0.0,0.0,1.0), 0.5).cpu()
    print(t, "cpu-formula-with-float-coefs vs oracle:", float((nxt-want).abs().max()), "| hip vs oracle:", float((h-want).abs().max()),
          "| e2: cpu", float((e2-e2_ref).abs().max()), "hip(with sb=0!)", float((e2_h-e).abs().max()))
    # e2 via hip with true sb: set sa=1, sap=0, sbp=1 -> out = e - sb*g*scale
    e2_h2 = engine.ddim_guided_step(x.to(dev), e.to(dev), gr.to(dev).reshape(1,-1), 1, (1.0,sb,0.0,1.0), 0.5).cpu()
    print("    e2 hip vs ref", float((e2_h2-e2_ref).abs().max()))
    x0_h = engine.ddim_guided_step(x.to(dev), e2_ref.to(dev), None, 0, (sa,sb,1.0,0.0), 0.0).cpu()
    x0_ref = ((x - (1-so.alphas_cumprod[t])**0.5*e2_ref)/so.alphas_cumprod[t]**0.5).clamp(-1,1)
    print("    x0 hip vs ref", float((x0_h-x0_ref).abs().max()))
