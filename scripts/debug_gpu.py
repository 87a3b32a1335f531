import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
torch.set_printoptions(precision=5, linewidth=200)
g = util.load("g3_dyn2d.npz")
nv = int(g["dims"][6])
sd = util.dyn2d_sd(g["seed"], nv)
f = lambda k: torch.from_numpy(g[k]).to(dev)
args_cpu = [torch.from_numpy(g[k]) for k in ("fwd_xc", "fwd_xo", "fwd_xp", "fwd_t", "fwd_obj")]
def ident(sd):
    sd2 = {k: v.clone() for k, v in sd.items()}
    for i in range(1, 8):
        sd2[f"linears.{3*i}.weight"] = torch.eye(256); sd2[f"linears.{3*i}.bias"] = torch.zeros(256)
    for i in range(8):
        sd2[f"linears.{3*i+1}.weight"] = torch.ones(256); sd2[f"linears.{3*i+1}.bias"] = torch.zeros(256)
        sd2[f"linears.{3*i+1}.running_mean"] = torch.zeros(256); sd2[f"linears.{3*i+1}.running_var"] = torch.ones(256) - 1e-5
    return sd2
def run(sdx, tag):
    dyn = engine.Dynamics(2, sdx, 14, 2 * nv)
    y = dyn.forward2d(f("fwd_xc"), f("fwd_xo"), f("fwd_xp"), f("fwd_t"), f("fwd_obj")).cpu()
    r = orc.dyn2d_forward(sdx, *args_cpu)
    print(tag, "rel", util.rel_l2(y, r)); print(y[:3]); print(r[:3])
# (a) known z1: zero first layer weight, b1 = ramp; trunk identity
sda = ident(sd)
sda["linears.0.weight"] = torch.zeros_like(sd["linears.0.weight"]); sda["linears.0.bias"] = torch.linspace(-1, 2, 256)
run(sda, "(a) const z1, identity trunk")
# (a2) selector output: Wout = e_5, e_100, e_255
sdb = {k: v.clone() for k, v in sda.items()}
w = torch.zeros(3, 256); w[0, 5] = 1; w[1, 100] = 1; w[2, 255] = 1
sdb["output.weight"] = w; sdb["output.bias"] = torch.zeros(3)
run(sdb, "(a2) const z1, identity, selector out")
# (b) real first layer, identity trunk, selector output
sdc = ident(sd); sdc["output.weight"] = w; sdc["output.bias"] = torch.zeros(3)
run(sdc, "(b) real z1, identity, selector")
# (b2) only gripper part of first layer
for name, sl in (("obj", slice(0, 256)), ("ctrl", slice(256, 512)), ("pose", slice(512, 539)), ("time", slice(539, 795))):
    sdd = {k: v.clone() for k, v in sdc.items()}
    W = torch.zeros_like(sd["linears.0.weight"]); W[:, sl] = sd["linears.0.weight"][:, sl]
    sdd["linears.0.weight"] = W
    run(sdd, "(b2) only " + name)
