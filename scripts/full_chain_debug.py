"""Where does a free-running full-size chain leave the reference's trajectory?  usage: python3 scripts/full_chain_debug.py full_d00"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dgdm_amd import _lib, engine, sampler, synth
from tests import util
from tests.test_gpu_fullgrid3d import load_chain
from tests.test_gpu_parity import sched, finger_l2
_lib.device_init(0)
dev = torch.device("cuda:0")
part = sys.argv[1]
g, objs, ids = load_chain(part)
c64 = np.load(os.path.join(util.GOLDEN, "g9_calls64.npz"))
B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
o, gain = str(g["opt_obj"]), float(g["gain"])
net = engine.Unet1d(util.unet_sd(g["unet_seed"]))
dyn = engine.Dynamics(3, synth.scale_output(util.dyn3d_sd(g["dyn3d_seed"]), gain), L)
gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, N, 512, max_objects=max(2, objs.shape[0]))
gd.set_objects(objs.to(dev))
s = sched(T, S)
noise = synth.synth_noise(0, B, L).to(dev)
ug = sampler.unguided_sample(net, s, noise)
forced = lambda: sampler.StartStream(N, 512, util.unpack_starts(g["starts"].astype(np.int64), g["start_lens"]))
tr = []
end = sampler.guided_chains(net, gd, s, 'point_3d', noise, [(ids[0], o)], unguided=ug, starts=forced(), trace=tr)[0].cpu()
print("trace entries per step:", len(tr), [len(t) for t in tr][:2])
tx = g["trace_x"]            # [S][B][L][1]: x at the START of step si (the reference's)
for si in range(S):
    # tr[si] = (eps, grad, x_after?) - print what exists
    e_ref, g_ref = torch.from_numpy(g["trace_eps"][si]), torch.from_numpy(g["trace_grad"][si])
    e_h, g_h = tr[si][0][0].cpu().reshape(B, L, 1), tr[si][1][0].cpu().reshape(B, L, 1)
    de = (e_h.double() - e_ref.double()).reshape(B, -1).norm(dim=1) / e_ref.double().reshape(B, -1).norm(dim=1)
    dg = (g_h.double() - g_ref.double()).reshape(B, -1).norm(dim=1) / g_ref.double().reshape(B, -1).norm(dim=1).clamp_min(1e-30)
    print(f"step {si}: eps rel err per finger max {float(de.max()):.2e} (finger {int(de.argmax())}), grad rel err max {float(dg.max()):.2e} (finger {int(dg.argmax())}), median {float(dg.median()):.2e}; "
          f"|grad| ref {float(g_ref.norm()):.3e}")
d_ref = (end.double() - torch.from_numpy(g["guided"]).double()).reshape(B, -1).norm(dim=1)
d_64 = (end.double() - torch.from_numpy(c64[f"{part}/chain"]).double()).reshape(B, -1).norm(dim=1)
r_64 = (torch.from_numpy(g["guided"]).double() - torch.from_numpy(c64[f"{part}/chain"]).double()).reshape(B, -1).norm(dim=1)
print("end point per finger: HIP vs reference", [float("%.1e" % v) for v in d_ref])
print("end point per finger: HIP vs float64  ", [float("%.1e" % v) for v in d_64])
print("end point per finger: ref vs float64  ", [float("%.1e" % v) for v in r_64])

# ---- the HIP chain's own sensitivity: eps x (1 + rel N(0,1)) per step, three seeds (step-by-step form of sampler.guided_chains)
st_all = forced()
sw, step_starts = sampler.draw_chain_starts(gd, [(ids[0], o)], S, st_all)
scale = sampler.classifier_scale('point_3d', o)
for seed, rel in ((0, 0.0), (1, 1e-6), (2, 1e-6), (3, 1e-6)):
    gen = torch.Generator().manual_seed(seed)
    x = noise.reshape(1, B, L).contiguous()
    for si, t in enumerate(s.timesteps):
        t = int(t)
        ts = torch.full((B,), t, dtype=torch.int32, device=dev)
        eps = net.forward(x[0].reshape(B, L, 1), ts).reshape(1, B, L)
        if rel:
            eps = eps * (1.0 + rel * torch.randn(eps.shape, generator=gen)).to(dev)
        gr = gd.grad(x, t, [engine.make_objective(o, ids[0])], None, step_starts[si].reshape(-1))
        x = engine.ddim_guided_step(x[0], eps[0], gr[0], 1, s.coefficients(t), scale).reshape(1, B, L)
    e = x[0].cpu().reshape(B, L, 1)
    d64 = (e.double() - torch.from_numpy(c64[f"{part}/chain"]).double()).reshape(B, -1).norm(dim=1)
    dh = (e.double() - end.double()).reshape(B, -1).norm(dim=1)
    print(f"HIP step-by-step, eps rel {rel:g} seed {seed}: vs float64 max {float(d64.max()):.2e} (finger {int(d64.argmax())}), finger 16: {float(d64[16]):.2e}; vs the unperturbed HIP chain max {float(dh.max()):.2e} (finger {int(dh.argmax())})")

# ---- is finger 16's divergence a kink of the finger-level encoder?  gripper_encoder.0 pre-activations (float64) along both trajectories
sd = synth.scale_output(util.dyn3d_sd(g["dyn3d_seed"]), gain)
W0, b0 = sd["gripper_encoder.0.weight"].double(), sd["gripper_encoder.0.bias"].double()
W2, b2 = sd["gripper_encoder.2.weight"].double(), sd["gripper_encoder.2.bias"].double()
x = noise.reshape(1, B, L).contiguous()
xs = []
for si, t in enumerate(s.timesteps):
    t = int(t)
    xs.append(x[0].cpu().double())
    ts = torch.full((B,), t, dtype=torch.int32, device=dev)
    eps = net.forward(x[0].reshape(B, L, 1), ts).reshape(1, B, L)
    gr = gd.grad(x, t, [engine.make_objective(o, ids[0])], None, step_starts[si].reshape(-1))
    x = engine.ddim_guided_step(x[0], eps[0], gr[0], 1, s.coefficients(t), scale).reshape(1, B, L)
f = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for si in range(S):
    xh, xr = xs[si][f], torch.from_numpy(tx[si]).double()[f, :, 0]
    hh, hr = W0 @ xh + b0, W0 @ xr + b0
    size = W0.abs() @ xr.abs() + b0.abs()
    flips = torch.nonzero((hh > 0) != (hr > 0)).reshape(-1).tolist()
    print(f"step {si}: finger {f}: |x_hip - x_ref| {float((xh - xr).norm()):.2e}; gripper_encoder.0 units with different sign: {flips} "
          f"(their pre-activations / size: HIP {[float('%.1e' % (hh[j] / size[j])) for j in flips]}, ref {[float('%.1e' % (hr[j] / size[j])) for j in flips]}); "
          f"smallest |pre-activation| / size on the reference's x: {float((hr.abs() / size).min()):.1e}")
