import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from dgdm_amd.scheduler import DDIMScheduler
from tests import util
_lib.device_init(0); dev = torch.device("cuda:0")
g=util.load("g6_chains.npz")
s=DDIMScheduler(num_train_timesteps=15); s.set_timesteps(5)
xs, es, gs = g["trace3d_rotate_x"], g["trace3d_rotate_eps"], g["trace3d_rotate_grad"]
for si,t in enumerate(s.timesteps):
    t=int(t)
    x = torch.from_numpy(xs[si]).to(dev)
    g_ref = torch.from_numpy(gs[si]).to(dev).reshape(1, -1)
    nxt = engine.ddim_guided_step(x, torch.from_numpy(es[si]).to(dev), g_ref, 1, s.coefficients(t), 0.5).cpu()
    want = xs[si + 1] if si + 1 < 5 else g["guided3d_rotate"]
    d = (nxt.numpy()-want)
    print(t, "max abs", np.abs(d).max(), "argmax", np.unravel_index(np.abs(d).argmax(), d.shape), "x0-ish val", want.reshape(-1)[np.abs(d).argmax()])
