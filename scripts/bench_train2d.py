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
    t0 = time.time()
    for _ in range(n):
        inp = t._inputs(data[0], data[1], data[2], data[3], data[4])
    torch.cuda.synchronize(); dh = (time.time() - t0) / n
    import ctypes as C
    from dgdm_amd._lib import lib, dptr, stream_ptr, check
    c, nz, sa, sb, tt, o, p, ob, sc, rr = inp
    pred = torch.empty((rows, 3), device="cuda"); loss = C.c_float()
    t0 = time.time()
    for _ in range(n):
        check(lib().dgdm_trainer2d_step(t._h, dptr(c), dptr(nz), dptr(sa), dptr(sb), dptr(tt), dptr(o), dptr(p), dptr(ob), dptr(sc), rows, 1e-4, 1, dptr(pred), C.byref(loss), stream_ptr()))
    torch.cuda.synchronize(); dk = (time.time() - t0) / n
    print(f"   host draws + uploads {dh * 1e3:.2f} ms, C-ABI step alone {dk * 1e3:.2f} ms", flush=True)
    K = [L, 256, 2 * nv, 256, 128, 256, 795] + [256] * 7
    flops = rows * (3 * 2 * 256 * sum(K) - 2 * 256 * (L + 2 * nv + 128 + 27) + 3 * 2 * 3 * 256)     # no input gradient below the first layers
    print(f"rows {rows}: {dt * 1e3:.2f} ms/step (host draws included), {flops / dt / 1e12:.1f} TFLOP/s = {flops / dt / 157.3e12:.2f} of the f32 MFMA peak", flush=True)
