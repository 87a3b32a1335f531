"""How far does the FLOAT64 chain of a full-size fixture move when its eps-net output is perturbed by 1e-6 relative (the size of any float32
eps-net's rounding error)?  No float32 path involved.  usage: python scripts/chain64_sensitivity.py full_d00 [rel]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from dgdm_amd import synth
from oracle import dgdm_oracle as orc, fast64
from tests import util
part = sys.argv[1]
rel = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
g = np.load(os.path.join(util.GOLDEN, f"g9_3d_{part}.npz"))
c64 = np.load(os.path.join(util.GOLDEN, "g9_calls64.npz"))[f"{part}/chain"]
B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
o, gain = str(g["opt_obj"]), float(g["gain"])
oi = int(g["obj"].reshape(-1)[0])
sd32 = synth.scale_output(synth.synth_state_dict(synth.dyn3d_spec(42), int(g["dyn3d_seed"])), gain)
usd = synth.synth_state_dict(synth.unet_spec(), int(g["unet_seed"]))
sch = orc.DDIM(T); sch.set_timesteps(S)
t0 = time.time()
tab = fast64.ObjectTables64(fast64._f64(sd32), synth.synth_object_3d(int(g["obj_seed0"]) + oi))
print("tables", time.time() - t0, flush=True)
calls = util.unpack_starts(g["starts"].astype(np.int64), g["start_lens"])
assert o != 'convergence'
noise = synth.synth_noise(0, B, L)
gen = torch.Generator().manual_seed(1)
real = orc.unet1d_forward
def pert(sd, x, ts):
    out = real(sd, x, ts)
    return out * (1.0 + rel * torch.randn(out.shape, generator=gen, dtype=out.dtype))
orc.unet1d_forward = pert
end = fast64.guided_chain(usd, sd32, sch, L, G, P, 512, noise, [tab], o, None, calls).numpy()
orc.unet1d_forward = real
d = np.sqrt(((end - c64) ** 2).reshape(B, -1).sum(1))
print(f"{part}: float64 chain vs float64 chain with eps x (1 + {rel:g} N(0,1)): per finger", [float('%.1e' % v) for v in d], "max", float(d.max()), "finger", int(d.argmax()), f"({time.time() - t0:.0f}s)")
np.savez_compressed(os.path.join(util.GOLDEN, f"g9_3d_{part}_eps64.npz"), guided=end, eps64_floor=np.float64(d.max()), per_finger=d, rel=np.float64(rel))
