"""eps-net f16x3 convolutions over the input's dynamic range: samples scaled by 1e-3 ... 1e3 and an all-zero sample, against the float64 oracle."""
import sys; sys.path.insert(0, '.')
import torch
from dgdm_amd import engine, synth, _lib
from oracle import dgdm_oracle as orc
_lib.device_init(0)
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(synth.unet_spec(), 7)
sd64 = {k: v.double() for k, v in sd.items()}
net, chain = engine.Unet1d(sd), engine.Unet1d(sd, contraction_dtype="f32_mfma")
for L in (42, 14):
    for scale in (0.0, 1e-3, 1.0, 1e3, 1e6):
        x = synth.synth_noise(3, 16, L) * scale
        ts = torch.randint(0, 15, (16,), generator=torch.Generator().manual_seed(1))
        ref = orc.unet1d_forward(sd64, x.double(), ts)
        e = [float((n.forward(x.to(dev), ts.to(dev).int()).cpu().double() - ref).norm() / ref.norm()) for n in (net, chain)]
        print(f"L={L} input x {scale:g}: f16x3 {e[0]:.2e}  float32 chain {e[1]:.2e}")
