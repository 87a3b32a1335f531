import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dgdm_amd import _lib, engine, synth
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
sd = util.unet_sd(11)
os.environ["DGDM_UNET_BATCHED_MIN"] = "0"; ps = engine.Unet1d(sd)
os.environ["DGDM_UNET_BATCHED_MIN"] = "1"; bt = engine.Unet1d(sd)
g = torch.Generator().manual_seed(5)
for L, B in ((42, 8), (42, 5), (14, 8), (42, 1024)):
    x = torch.randn((B, L, 1), generator=g).to(dev); t = torch.randint(0, 15, (B,), generator=g).to(dev)
    a = ps.forward(x, t); torch.cuda.synchronize(); print("per-sample ok", L, B, flush=True)
    b = bt.forward(x, t); torch.cuda.synchronize(); print("batched ok", L, B, bool(torch.equal(a, b)), float((a - b).abs().max()), flush=True)
