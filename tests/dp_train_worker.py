"""Worker of tests/test_gpu_train.py::test_trainer2d_data_parallel: one rank of a data-parallel Trainer run (launched under
torch.distributed.run).  argv: data seed, rows, steps, output .npz (written by rank 0)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_amd import dist as ddist  # noqa: E402
from dgdm_amd import _lib, synth  # noqa: E402
from tests.test_gpu_train import _args, dp_data  # noqa: E402


def main():
    seed, rows, steps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    world, rank, local = ddist.init_from_env()
    _lib.device_init(local)
    from dynamics.trainer import Trainer
    sd, data = dp_data(seed, rows)
    t = Trainer(_args(0.0))
    t.create_model(state_dict=sd)
    ddist.sync_start_stream_seed()          # building the nn.Module drew its initial weights from the CPU generator: back to the pinned seed
    rec = {}
    for k in range(steps):
        loss, pred = t.step(*data)
        rec[f"loss{k}"], rec[f"pred{k}"] = np.float64(loss), pred.cpu().numpy()
    pi, li = t.inference(*data)
    rec["inf_pred"], rec["inf_loss"] = pi.cpu().numpy(), np.float64(li)
    for k, v in t.state_dict().items():
        rec["sd/" + k] = v.numpy()
    for k, v in t.gradients().items():
        rec["grad/" + k] = v.numpy()
    if rank == 0:
        np.savez(out, world=np.int64(world), **rec)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
