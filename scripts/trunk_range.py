"""f16x3 trunk over the pre-activations' dynamic range: the 3-D first-layer weights (and bias) scaled by s, the second layer's by 1 / s -
layer 1's outputs, whose row scale for layer 2 comes from a BOUND, move over 12 orders of magnitude; cond_fn gradient against the float32 MFMA chain."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
g = np.load("tests/golden/g9_3d_rotate.npz")
B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
for s in (1e-6, 1e-3, 1.0, 1e3, 1e6):
    sd = dict(synth.scale_output(util.dyn3d_sd(g["dyn3d_seed"]), float(g["gain"])))
    sd = {k: v.clone() for k, v in sd.items()}
    for k in ("linears.0.weight", "linears.0.bias", "linears.1.running_mean"):
        sd[k] = sd[k] * s
    sd["linears.1.running_var"] = sd["linears.1.running_var"] * s * s
    # BatchNorm (eval) divides the scale out again; scale its affine output instead, and undo it in the next layer's input weights
    sd["linears.1.weight"] = sd["linears.1.weight"] * s
    sd["linears.1.bias"] = sd["linears.1.bias"] * s
    sd["linears.3.weight"] = sd["linears.3.weight"] / s
    dyn = engine.Dynamics(3, sd, L)
    res = {}
    for mode in ("f32_mfma", "f32"):
        gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, N, 512, max_objects=2, contraction_dtype=mode)
        gd.set_objects(torch.from_numpy(g["objs"]).to(dev))
        st = sampler.StartStream(N, 512, util.unpack_starts(g["starts"].astype(np.int64), g["start_lens"]))
        x = torch.from_numpy(g["trace_x"][0]).to(dev).reshape(1, B, L)
        res[mode] = gd.grad(x, 12, [engine.make_objective("rotate", 0)], None, st.call(gd.rows)).cpu().double()
    d = float((res["f32"] - res["f32_mfma"]).norm() / res["f32_mfma"].norm())
    print(f"layer-1 scale {s:g}: |grad| {float(res['f32_mfma'].norm()):.3e}, f16x3 vs float32 chain {d:.2e}, finite {bool(torch.isfinite(res['f32']).all())}")
