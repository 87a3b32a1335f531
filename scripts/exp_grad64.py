"""Is a 1e-4-level gradient difference between the HIP path and the float32 oracle noise or error?  Compare both with a float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0); dev = torch.device("cuda:0")
torch.set_num_threads(32)
B, G, P, L, T, S, sub = 2, 16, 3, 42, 15, 5, 64
dsd = util.dyn3d_sd(33)
d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in dsd.items()}
obj = synth.synth_object_3d(5)
dyn = engine.Dynamics(3, dsd, L)
gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 1, T, 512, sub, 1); gd.set_objects(obj[None].to(dev))
s32 = util.setup('point_3d', None, dsd, T, S, L, G, P, sub)
s64 = util.setup('point_3d', None, d64, T, S, L, G, P, sub)
x = synth.synth_noise(0, B, L)
t = torch.full((B,), 12, dtype=torch.int64)
for o in ('shift_up', 'rotate'):
    torch.manual_seed(1); st = sampler.StartStream(512, sub).call(gd.rows)
    gh = gd.grad(x.reshape(1, B, L).to(dev), 12, [engine.make_objective(o, 0)], None, st).cpu().reshape(B, L, 1).double()
    lens = [n for r0 in range(0, gd.rows, sub) for n in (min(sub, gd.rows - r0),) * 2]
    g32 = orc.cond_fn(s32, x, t, o, obj, (-1.0, 1.0), None, orc.StartLog(util.unpack_starts(st, lens))).double()
    g64 = orc.cond_fn(s64, x.double(), t, o, obj.double(), (-1.0, 1.0), None, orc.StartLog(util.unpack_starts(st, lens)))
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(o, "HIP vs f64:", rel(gh, g64), " torch-f32 vs f64:", rel(g32, g64), " HIP vs torch-f32:", rel(gh, g32),
          " per finger HIP-f64:", [rel(gh[i], g64[i]) for i in range(B)], " torch32-f64:", [rel(g32[i], g64[i]) for i in range(B)])
print("fast_ok:", gd.debug_fps_path(True))
for o in ('shift_up',):
    torch.manual_seed(1); st = sampler.StartStream(512, sub).call(gd.rows)
    gs = gd.grad(x.reshape(1, B, L).to(dev), 12, [engine.make_objective(o, 0)], None, st).cpu().reshape(B, L, 1).double()
    lens = [n for r0 in range(0, gd.rows, sub) for n in (min(sub, gd.rows - r0),) * 2]
    g32 = orc.cond_fn(s32, x, t, o, obj, (-1.0, 1.0), None, orc.StartLog(util.unpack_starts(st, lens))).double()
    print("forced per-row FPS: HIP vs torch-f32 per finger:", [float((gs[i]-g32[i]).norm()/g32[i].norm()) for i in range(B)])
