"""eps-net forward in the three arithmetic modes against the float64 oracle, and their time per call."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, time
from dgdm_amd import engine, synth, _lib
from oracle import dgdm_oracle as orc
from tests import util
_lib.device_init(0)
dev = torch.device("cuda:0")
for L in (42, 14):
    sd = synth.synth_state_dict(synth.unet_spec(), 7)
    sd64 = {k: v.double() for k, v in sd.items()}
    B = 64
    x = synth.synth_noise(3, B, L)
    ts = torch.randint(0, 15, (B,), generator=torch.Generator().manual_seed(1))
    ref64 = orc.unet1d_forward(sd64, x.double(), ts)
    ref32 = orc.unet1d_forward(sd, x, ts)
    print(f"L={L} oracle f32 vs f64: {float((ref32.double() - ref64).norm() / ref64.norm()):.2e}")
    for mode in ("f32_mfma", "f32", "bf16"):
        net = engine.Unet1d(sd, contraction_dtype=mode)
        out = net.forward(x.to(dev), ts.to(dev).int()).cpu().double()
        big = synth.synth_noise(4, 1024, L).to(dev); tb = torch.randint(0, 15, (1024,)).int().to(dev)
        for _ in range(3): net.forward(big, tb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): net.forward(big, tb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"  {mode:9s} vs f64: {float((out - ref64).norm() / ref64.norm()):.2e}   max {float((out - ref64).abs().max() / ref64.abs().max()):.2e}   {dt * 1e3:.3f} ms per 1024 samples")
