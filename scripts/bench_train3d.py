"""Times Trainer.step of the 3-D dynamics model (csrc/train3d.hip): ms per step and rows/s.  python scripts/bench_train3d.py [rows ...]
(under `rocprofv3 --kernel-trace --stats` for the kernel breakdown quoted in DESIGN.md 4.6)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_train3d import _args
from tests import util
from dgdm_amd import _lib, synth
from dgdm_amd.dynamics.trainer import Trainer
_lib.device_init(0)
sd = util.dyn3d_sd(57)
for rows in [int(a) for a in sys.argv[1:]] or [512]:
    rs = np.random.RandomState(0)
    ctrl = torch.from_numpy(rs.uniform(-1, 1, (rows, 3, 42)).astype(np.float32))
    obj = torch.stack([synth.synth_object_3d(80 + i % 16) for i in range(rows)]).permute(0, 2, 1).contiguous()
    ori = torch.from_numpy(rs.uniform(-1, 1, (rows, 1)).astype(np.float32))
    pos = torch.from_numpy(rs.uniform(-1, 1, (rows, 2)).astype(np.float32))
    score = torch.from_numpy(rs.normal(0, 1, (rows, 3)).astype(np.float32))
    data = [x.cuda() for x in (ctrl, score, ori, pos, obj)]
    t = Trainer(_args(False, 0.0)); t.create_model(sd)
    for _ in range(2):
        t.step(*data)
    torch.cuda.synchronize(); t0 = time.time()
    n = 5
    for _ in range(n):
        t.step(*data)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"rows {rows}: {dt * 1e3:.2f} ms/step, {rows / dt:.0f} rows/s", flush=True)
