"""f32_f16x3 trunk against the six-product form and the oracle: golden cond_fn gradients (2-D g3, 3-D g5), full-grid first-step gradients vs float64."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, time
from dgdm_amd import engine, sampler, synth, _lib
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
# 3-D full grid, first call of g9 'rotate' vs float64
c64 = np.load("tests/golden/g9_calls64.npz")
for part in ("rotate", "convergence_b", "shift_left_b", "ccw_down"):
    g = np.load(f"tests/golden/g9_3d_{part}.npz")
    B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
    o, gain = str(g["opt_obj"]), float(g["gain"]); oi = int(g["obj"]) if "obj" in g.files else 0
    dyn = engine.Dynamics(3, synth.scale_output(util.dyn3d_sd(g["dyn3d_seed"]), gain), L)
    res = {}
    for mode in ("f32_f16x3", "f32_mfma"):
        gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, N, 512, max_objects=2, contraction_dtype=mode)
        gd.set_objects(torch.from_numpy(g["objs"]).to(dev))
        st = sampler.StartStream(N, 512, util.unpack_starts(g["starts"].astype(np.int64), g["start_lens"]))
        rc = None
        if o == 'convergence':
            net = engine.Unet1d(util.unet_sd(g["unet_seed"]))
            from tests.test_gpu_parity import sched
            s = sched(T, S)
            ug = sampler.unguided_sample(net, s, synth.synth_noise(0, B, L).to(dev))
            centers = sampler.convergence_centers(gd, 'point_3d', ug, [oi], st.call(gd.sweep_rows))
            rc = torch.from_numpy(gd.rowcoef(centers[0])).to(dev).reshape(1, -1)
        errs = []
        for si in range(S):
            x = torch.from_numpy(g["trace_x"][si]).to(dev).reshape(1, B, L)
            tt = [12, 9, 6, 3, 0][si]
            gr = gd.grad(x, tt, [engine.make_objective(o, oi)], rc, st.call(gd.rows)).cpu().double().reshape(B, L).numpy()
            errs.append(float(np.linalg.norm(gr - c64[f"{part}/grad"][si]) / np.linalg.norm(c64[f"{part}/grad"][si])))
        res[mode] = errs
    print(part, {k: [float('%.1e' % e) for e in v] for k, v in res.items()})
# timing: 3-D 32 pairs, 2-D 4 pairs
import subprocess
for wl in ("3d", "2d"):
    for mode in ("f32_f16x3",):
        pass
