"""Host-side cost of one 3-D bench step: FPS start draws (background thread in bench.py) and the host part of guidance_grad."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgdm_amd import engine, sampler, synth, _lib
import bench
_lib.device_init(0)
dev = torch.device("cuda:0")
wl = bench.Workload("3d", 32, dev, 0, sys.argv[1] if len(sys.argv) > 1 else "bf16")
t0 = time.perf_counter(); pre = wl.draw(0); t1 = time.perf_counter()
print("draw_chain_starts for one step: %.1f ms (threads=%d)" % ((t1 - t0) * 1e3, torch.get_num_threads()))
objs = wl.objects(0)
wl.guid.set_objects(objs)
x = wl.noise.reshape(1, wl.B, wl.L).expand(32, -1, -1).contiguous()
objectives = [engine.make_objective(o, oi) for oi, o in wl.chains(0)]
st = pre[1][0].reshape(-1)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    g = wl.guid.grad(x, 6, objectives, None, st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("guidance_grad: host returns after %.2f ms, GPU done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
