import sys; sys.path.insert(0, '.')
import time, torch, argparse, contextlib, io, cProfile, pstats
from dgdm_amd import synth, _lib
from dgdm_amd.dynamics.trainer import Trainer
_lib.device_init(0)
dev = torch.device("cuda:0")
L, nv, T, rows = 14, 100, 15, 128 * 9000
args = argparse.Namespace(use_sub_batch=False, sub_bs=1024, grid_size=360, learning_rate=1e-4, weight_decay=0.0, num_epochs=100,
                          checkpoint_path=None, fingers_3d=False, ctrlpts_dim=L, object_max_num_vertices=nv, num_timesteps_per_batch=1,
                          num_inference_steps=5, num_train_timesteps=T)
sd = synth.synth_state_dict(synth.dyn2d_spec(L, 2 * nv), 41)
tr = Trainer(args)
with contextlib.redirect_stdout(io.StringIO()):
    tr.create_model(state_dict=sd)
g = torch.Generator().manual_seed(3)
data = [torch.rand(shape, generator=g) * 2 - 1 for shape in ((rows, L), (rows, 3), (rows, 1), (rows, 2), (rows, 2 * nv))]
data = [d.to(dev) for d in data]
for _ in range(2):
    tr.step(*data, rows_per_sample=9000)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(4):
    tr.step(*data, rows_per_sample=9000)
torch.cuda.synchronize()
pr.disable()
print("wall per step", (time.perf_counter() - t0) / 4)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
