import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dgdm_amd import _lib, engine, synth
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0); dev = torch.device('cuda:0')
nv, B, G, P, L, T = 100, 5, 7, 3, 14, 15
sd = util.dyn2d_sd(77, nv)
dyn = engine.Dynamics(2, sd, L, 2 * nv)
objs = [synth.synth_object_2d(i, nv) for i in range(3)]
s = util.setup('point', None, sd, T, 5, L, G, P)
chains = [(0, 'rotate'), (1, 'shift_left'), (2, 'clockwise_up'), (1, 'rotate'), (0, 'convergence'), (2, 'rotate_counterclockwise')]
xs = torch.stack([synth.synth_noise(50 + i, B, L).clamp(-1, 1) for i in range(len(chains))])
centers = torch.tensor([2, 0, 6, 3, 1])
refs = [orc.cond_fn(s, xs[c], torch.full((B,), 6, dtype=torch.int64), o, objs[oi], (-1.0, 1.0), centers if o == 'convergence' else None) for c, (oi, o) in enumerate(chains)]
s64 = util.setup('point', None, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, T, 5, L, G, P)
refs64 = [orc.cond_fn(s64, xs[c].double(), torch.full((B,), 6, dtype=torch.int64), o, objs[oi].double(), (-1.0, 1.0), centers if o == 'convergence' else None) for c, (oi, o) in enumerate(chains)]
for mode in ('f32', 'f32_mfma'):
    gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 6, T, nv, 0, max_objects=3, contraction_dtype=mode)
    gd.set_objects(torch.stack(objs).to(dev))
    rc = np.zeros((len(chains), gd.rows), np.float32); rc[4] = gd.rowcoef(centers)
    gr = gd.grad(xs.reshape(len(chains), B, L).to(dev), 6, [engine.make_objective(o, oi) for oi, o in chains], torch.from_numpy(rc).to(dev)).cpu()
    print(mode, 'vs f32 oracle', ['%.1e' % util.rel_l2(gr[c].reshape(B, L, 1), refs[c]) for c in range(6)], '| vs f64 oracle', ['%.1e' % util.rel_l2(gr[c].reshape(B, L, 1), refs64[c]) for c in range(6)])
print('f32 oracle vs f64 oracle', ['%.1e' % util.rel_l2(refs[c], refs64[c]) for c in range(6)])
