"""bf16-contraction trunk vs the oracle's bf16 emulation and vs the f32 oracle (2-D and 3-D), error printout."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "2d"
if which == "2d":
    B, G, P, L, T, nv = 5, 24, 3, 14, 15, 100       # C = 216 cells -> 7 tiles per finger (odd), 35 tiles per chain
    sd = util.dyn2d_sd(22, nv)
    dyn = engine.Dynamics(2, sd, L, 2 * nv)
    objs = [synth.synth_object_2d(i, nv) for i in range(2)]
    gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 4, T, nv, 0, max_objects=2)
    gd.set_objects(torch.stack(objs).to(dev))
    chains = [(0, 'rotate'), (1, 'shift_left'), (0, 'counterclockwise_up')]
    x = torch.stack([synth.synth_noise(70 + i, B, L) for i in range(len(chains))]).clamp(-1, 1)
    s = util.setup('point', None, sd, T, 5, L, G, P)
    for dt in ("f32", "bf16"):
        gd.set_contraction_dtype(dt)
        got = gd.grad(x.reshape(len(chains), B, L).to(dev), 6, [engine.make_objective(o, oi) for oi, o in chains], None, None).cpu()
        for c, (oi, o) in enumerate(chains):
            ref32 = orc.cond_fn(s, x[c], torch.full((B,), 6, dtype=torch.int64), o, objs[oi])
            with orc.contraction('bf16'):
                ref16 = orc.cond_fn(s, x[c], torch.full((B,), 6, dtype=torch.int64), o, objs[oi])
            print(dt, o, "vs f32 oracle %.3e  vs bf16 oracle %.3e   (oracle bf16 vs f32 %.3e)" % (
                util.rel_l2(got[c].reshape(B, L, 1), ref32), util.rel_l2(got[c].reshape(B, L, 1), ref16), util.rel_l2(ref16, ref32)))
else:
    B, G, P, L, T, sub = 3, 4, 3, 42, 15, 11         # C = 36 -> 2 tiles per finger
    sd = util.dyn3d_sd(44)
    dyn = engine.Dynamics(3, sd, L)
    objs = torch.stack([synth.synth_object_3d(31), synth.synth_object_3d(32)])
    gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, 512, sub, max_objects=2)
    gd.set_objects(objs.to(dev))
    x = torch.stack([synth.synth_noise(60, B, L), synth.synth_noise(61, B, L)]).clamp(-1, 1)
    torch.manual_seed(3)
    st = sampler.StartStream(512, sub)
    starts = np.concatenate([st.call(gd.rows), st.call(gd.rows)])
    chains = ((0, 'rotate'), (1, 'counterclockwise_left'))
    s = util.setup('point_3d', None, sd, T, 5, L, G, P, sub)
    for dt in ("f32", "bf16"):
        gd.set_contraction_dtype(dt)
        got = gd.grad(x.reshape(2, B, L).to(dev), 3, [engine.make_objective(o, oi) for oi, o in chains], None, starts).cpu()
        for c, (oi, o) in enumerate(chains):
            def log():
                return orc.StartLog(util.unpack_starts(starts[c * 2 * gd.rows:(c + 1) * 2 * gd.rows],
                                                       [n for r0 in range(0, gd.rows, sub) for n in (min(sub, gd.rows - r0),) * 2]))
            ref32 = orc.cond_fn(s, x[c], torch.full((B,), 3, dtype=torch.int64), o, objs[oi], (-1.0, 1.0), None, log())
            with orc.contraction('bf16'):
                ref16 = orc.cond_fn(s, x[c], torch.full((B,), 3, dtype=torch.int64), o, objs[oi], (-1.0, 1.0), None, log())
            print(dt, o, "vs f32 oracle %.3e  vs bf16 oracle %.3e   (oracle bf16 vs f32 %.3e)" % (
                util.rel_l2(got[c].reshape(B, L, 1), ref32), util.rel_l2(got[c].reshape(B, L, 1), ref16), util.rel_l2(ref16, ref32)))
