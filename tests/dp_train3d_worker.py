"""Worker of tests/test_gpu_train3d.py::test_trainer3d_data_parallel: one rank of a data-parallel 3-D Trainer step (launched under
torch.distributed.run).  argv: rows, output .npz (written by rank 0)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_amd import dist as ddist  # noqa: E402
from dgdm_amd import _lib  # noqa: E402
from tests.test_gpu_train3d import _args, dp3d_data  # noqa: E402


def main():
    rows, out = int(sys.argv[1]), sys.argv[2]
    world, rank, local = ddist.init_from_env()
    _lib.device_init(local)
    from dynamics.trainer import Trainer
    sd, data = dp3d_data(rows)
    t = Trainer(_args(False, 0.0))
    t.create_model(state_dict=sd)
    ddist.sync_start_stream_seed()          # building the nn.Module drew its initial weights from the CPU generator: back to the pinned seed
    rec = {}
    loss, pred = t.step(*data)
    rec["loss0"], rec["pred0"] = np.float64(loss), pred.cpu().numpy()
    for k, v in t.gradients().items():
        rec["grad/" + k] = v.numpy()
    for k, v in t.state_dict().items():
        rec["sd/" + k] = v.numpy()
    pi, li = t.inference(*data)
    rec["inf_pred"], rec["inf_loss"] = pi.cpu().numpy(), np.float64(li)
    # every rank must hold the same parameters after the step: checksum gathered on rank 0
    flat = torch.cat([v.reshape(-1).double() for k, v in sorted(t.state_dict().items()) if v.is_floating_point()])
    sums = ddist.all_gather_rows(torch.stack([flat.sum(), flat.abs().sum()]).cuda()).cpu().numpy()
    rec["replica_sums"] = sums
    if rank == 0:
        np.savez(out, world=np.int64(world), **rec)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
