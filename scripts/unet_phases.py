#!/usr/bin/env python3
"""Mean shader-clock time per phase of unet_kernel (experiment build: DGDM_EXTRA_FLAGS=-DDGDM_UNET_CLOCKS python -m dgdm_amd.build --force).
Run on the GPU box: python scripts/unet_phases.py [bf16]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_amd import _lib, engine, synth  # noqa: E402

_lib.device_init(0)
dev = torch.device("cuda:0")
net = engine.Unet1d(synth.synth_state_dict(synth.unet_spec(), 11), contraction_dtype="bf16" if "bf16" in sys.argv else "f32")
x = synth.synth_noise(0, 1024, 42).to(dev)
t = torch.full((1024,), 9, device=dev)
for _ in range(3):
    net.forward(x, t)
torch.cuda.synchronize()
n = 1024 * 64
buf = (C.c_longlong * n)()
fn = _lib.lib().dgdm_debug_unet_clocks
fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
assert fn(buf, n) == 0
c = np.frombuffer(buf, dtype=np.int64).reshape(1024, 64)
nst = int((c[0] != 0).sum())
d = np.diff(c[:, :nst], axis=1).astype(np.float64)
names = ["step encoder"]
for b in range(8):
    names += [f"res{b} conv0", f"res{b} gn0", f"res{b} conv1", f"res{b} gn1", f"res{b} residual"]
# kernel-level stamps interleave: res0,res1, downsample, res2..res5, concat, res6, res7, upsample, final conv, final gn, out
order = ["step encoder"] + names[1:11] + ["downsample"] + names[11:31] + ["concat"] + names[31:41] + ["upsample", "final conv", "final gn", "output conv"]
tot = d.sum(1).mean()
print(f"stamps {nst}, mean cycles per workgroup {tot:.0f}")
agg = {}
for i in range(d.shape[1]):
    nm = order[i] if i < len(order) else f"phase{i}"
    kind = nm.split()[-1] if nm.startswith("res") else nm
    agg[kind] = agg.get(kind, 0.0) + d[:, i].mean()
    print(f"  {nm:16s} {d[:, i].mean():9.0f} cycles  {100 * d[:, i].mean() / tot:5.1f} %")
print({k: f"{100 * v / tot:.1f} %" for k, v in agg.items()})
