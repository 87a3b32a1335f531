import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dgdm_amd import engine, synth, _lib
_lib.device_init(0)
sd = synth.synth_state_dict(synth.unet_spec(), 5)
for B, L in ((1024, 42), (2048, 14)):
    tr = engine.UnetTrainer(sd, L)
    rs = np.random.RandomState(0)
    x0 = torch.from_numpy(rs.uniform(-1, 1, (B, L, 1)).astype(np.float32)).cuda()
    noise = torch.randn(B, L, 1).cuda(); ts = torch.randint(0, 15, (B,)).cuda()
    sa = torch.rand(B).cuda(); sb = (1 - sa * sa).sqrt()
    for _ in range(3): tr.step(x0, noise, sa, sb, ts, 1e-4)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): tr.step(x0, noise, sa, sb, ts, 1e-4, want_loss=False)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"B={B} L={L}: {dt*1e3:.2f} ms/step, {B/dt:.0f} samples/s")
