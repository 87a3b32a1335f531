import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import _lib, engine, synth
_lib.device_init(0)
dev = torch.device("cuda:0")
dyn = engine.Dynamics(3, synth.synth_state_dict(synth.dyn3d_spec(42), 33), 42)
n = 16
gd = engine.Guidance(dyn, 2, 2, 1, (-1.0, 1.0), 1, 15, 512, 512, max_objects=n)
objs = torch.stack([synth.synth_object_3d(i) for i in range(n)])
gd.set_objects(objs.to(dev))
print("fast_ok per object:", gd.debug_fps_path(False))
# count in-radius neighbours (r=0.4) to see how crowded balls are
for i in range(4):
    p = objs[i]
    d = ((p[:, None] - p[None]) ** 2).sum(-1)
    cnt = (d <= 0.16).sum(1)
    print("obj", i, "nbrs r=0.4: min/median/max", int(cnt.min()), int(cnt.median()), int(cnt.max()), "frac>64:", float((cnt > 64).float().mean()))
