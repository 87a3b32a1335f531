"""Dumps a few 2-D / 3-D guidance gradients of the default trunk (for bit comparisons between two builds: scripts/ab_bits.sh)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dgdm_amd import _lib, engine, sampler, synth
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
out = {}
g = np.load(os.path.join(util.GOLDEN, "g9_3d_rotate.npz"))
B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
dyn = engine.Dynamics(3, synth.scale_output(util.dyn3d_sd(g["dyn3d_seed"]), float(g["gain"])), L)
gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, N, 512, max_objects=2)
gd.set_objects(torch.from_numpy(g["objs"]).to(dev))
st = sampler.StartStream(N, 512, util.unpack_starts(g["starts"].astype(np.int64), g["start_lens"]))
for si in range(3):
    x = torch.from_numpy(g["trace_x"][si]).to(dev).reshape(1, B, L)
    out[f"g3_{si}"] = gd.grad(x, 12 - 3 * si, [engine.make_objective("rotate", 0)], None, st.call(gd.rows)).cpu().numpy()
nv = 100
dyn2 = engine.Dynamics(2, util.dyn2d_sd(77, nv), 14, 2 * nv)
gd2 = engine.Guidance(dyn2, 5, 360, 5, (-1.0, 1.0), 1, 15, nv, 0, max_objects=1)
gd2.set_objects(synth.synth_object_2d(1, nv)[None].to(dev))
x2 = synth.synth_noise(50, 5, 14).clamp(-1, 1).reshape(1, 5, 14).to(dev)
for o in ("rotate", "shift_left"):
    out["g2_" + o] = gd2.grad(x2, 6, [engine.make_objective(o, 0)], None).cpu().numpy()
np.savez(sys.argv[1], **out)
