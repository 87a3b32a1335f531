"""Times Trainer.step of the 2-D dynamics model on the GPU (csrc/train2d.hip): ms per step and float32 MFMA rate.
python scripts/bench_train2d.py [rows ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_train import _args
from dgdm_amd import _lib, synth
from dynamics.trainer import Trainer
_lib.device_init(0)
L, nv = 14, 100
sd = synth.synth_state_dict(synth.dyn2d_spec(L, 2 * nv), 41)
for rows in [int(a) for a in sys.argv[1:]] or [46080, 368640]:
    rs = np.random.RandomState(0)
    data = [torch.from_numpy(rs.uniform(-1, 1, s).astype(np.float32)).cuda() for s in ((rows, L), (rows, 3), (rows, 1), (rows, 2), (rows, 2 * nv))]
    t = Trainer(_args(0.0, L, nv)); t.create_model(state_dict=sd)
    for _ in range(3):
        t.step(*data)
    torch.cuda.synchronize(); t0 = time.time()
    n = 10
    for _ in range(n):
        t.step(*data)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    K = [L, 256, 2 * nv, 256, 128, 256, 795] + [256] * 7
    flops = rows * (3 * 2 * 256 * sum(K) - 2 * 256 * (L + 2 * nv + 128 + 27) + 3 * 2 * 3 * 256)     # no input gradient below the first layers
    print(f"rows {rows}: {dt * 1e3:.2f} ms/step (host draws included), {flops / dt / 1e12:.1f} TFLOP/s = {flops / dt / 157.3e12:.2f} of the f32 MFMA peak", flush=True)
