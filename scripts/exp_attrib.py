"""Which stage makes a free-running 3-D chain part from the reference's?  (MI355X; tests/golden/g9_3d_<part>.npz)

Runs the chain of tests/test_gpu_fullgrid.py::test_fullgrid_3d four times with the eps-net output and the guidance gradient each
taken either from the HIP path (evaluated at the chain's own x) or from the reference's recorded trace (valid while x stays within
rounding of the recorded x), and prints the finger-L2 distance to the reference's x after every step.

    python scripts/exp_attrib.py [part ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgdm_amd import _lib, engine, sampler, synth           # noqa: E402
from dgdm_amd.scheduler import DDIMScheduler                # noqa: E402
from tests import util                                      # noqa: E402

_lib.device_init(0)
dev = torch.device("cuda:0")


def fl2(a, b):
    return float(util.finger_err(a, b).max())


for part in (sys.argv[1:] or ["convergence", "rotate"]):
    g = np.load(os.path.join(util.GOLDEN, f"g9_3d_{part}.npz"))
    B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
    o, gain = str(g["opt_obj"]), float(g["gain"])
    oi = int(g["obj"]) if "obj" in g.files else 0
    net = engine.Unet1d(util.unet_sd(g["unet_seed"]))
    dyn = engine.Dynamics(3, synth.scale_output(util.dyn3d_sd(g["dyn3d_seed"]), gain), L)
    gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, N, 512, max_objects=2)
    gd.set_objects(torch.from_numpy(g["objs"]).to(dev))
    s = DDIMScheduler(num_train_timesteps=T)
    s.set_timesteps(S)
    noise = synth.synth_noise(0, B, L).to(dev)
    ug = sampler.unguided_sample(net, s, noise)
    multi = part.startswith("multi")
    objs = [0, 1] if multi else [oi]
    forced = lambda: sampler.StartStream(N, 512, util.unpack_starts(g["starts"].astype(np.int64), g["start_lens"]))      # noqa: E731
    st = forced()
    rc = None
    if o == 'convergence':
        centers = sampler.convergence_centers(gd, 'point_3d', ug, [oi], st.call(gd.sweep_rows))
        rc = torch.from_numpy(gd.rowcoef(centers[0])).to(dev).reshape(1, -1)
    steps = [np.concatenate([st.call(gd.rows) for _ in objs]) for _ in range(S)]
    scale = sampler.classifier_scale('point_3d', o, multi=multi)
    xs, es, gs = g["trace_x"], g["trace_eps"], g["trace_grad"]
    n = len(objs)
    if part == (sys.argv[1:] or ["convergence"])[0]:
        # the eps-net alone: HIP and the reference's recorded output against a float64 evaluation (oracle, CPU) on the recorded inputs
        from oracle import dgdm_oracle as orc
        usd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in util.unet_sd(g["unet_seed"]).items()}
        for si, t in enumerate(s.timesteps):
            ts = torch.full((B,), int(t), dtype=torch.int64)
            with torch.no_grad():
                e64 = orc.unet1d_forward(usd64, torch.from_numpy(xs[si]).double(), ts)
            eh = net.forward(torch.from_numpy(xs[si]).to(dev), ts.to(dev).int()).cpu()
            print(f"  eps-net step {si}: rel error vs float64: reference {util.rel_l2(es[si], e64):.2e}  HIP {util.rel_l2(eh, e64):.2e}  | HIP vs reference {util.rel_l2(eh, es[si]):.2e}")
    print(f"== {part}: {o}, gain {gain}, scale {scale}; |eps| finger L2 ~ {float(np.linalg.norm(es[0].reshape(B, -1), axis=1).mean()):.2f}, "
          f"|sb*scale*grad| ~ {[float('%.2g' % (s.coefficients(int(t))[1] * scale * np.linalg.norm(gs[i * n].reshape(B, -1), axis=1).mean())) for i, t in enumerate(s.timesteps)]}")
    for eps_src, grad_src in (("hip", "hip"), ("ref", "hip"), ("hip", "ref"), ("ref", "ref")):
        x = noise.reshape(B, L).clone()
        dev_x = []
        for si, t in enumerate(s.timesteps):
            t = int(t)
            dev_x.append(fl2(x.cpu().reshape(B, L, 1), xs[si]))
            if eps_src == "hip":
                eps = net.forward(x.reshape(B, L, 1), torch.full((B,), t, device=dev, dtype=torch.int32)).reshape(B, L)
            else:
                eps = torch.from_numpy(es[si]).to(dev).reshape(B, L)
            if grad_src == "hip":
                gr = gd.grad(x.reshape(1, B, L).expand(n, -1, -1).contiguous(), t, [engine.make_objective(o, k) for k in objs], rc, steps[si])
            else:
                gr = torch.from_numpy(gs[si * n:(si + 1) * n]).to(dev).reshape(n, B, L)
            x = engine.ddim_guided_step(x, eps, gr.reshape(n, B, L), n, s.coefficients(t), scale)
        end = fl2(x.cpu().reshape(B, L, 1), g["guided"])
        print(f"  eps {eps_src} grad {grad_src}: x deviation before each step {[float('%.2g' % v) for v in dev_x]} -> end point {end:.3g}")
