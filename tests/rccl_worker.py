"""Worker of tests/test_gpu_dist.py: a process group of ONE rank on the "nccl" backend (= RCCL on ROCm) - the only way a 1-GPU box
can load librccl and run the device branches of dgdm_amd/dist.py; prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgdm_amd import dist as ddist      # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    x = torch.arange(5 * 2 * 42, dtype=torch.float32, device=dev).reshape(5, 2, 42, 1)
    g = ddist.gather_pairs(x, 5)
    h = ddist.gather_pairs(x, 5, async_op=True)                 # the asynchronous form: RCCL's stream, completed by wait()
    res["gather_pairs"] = bool(g.is_cuda and torch.equal(g, x) and torch.equal(h.wait(), x))
    t = torch.linspace(-1, 1, 1200, device=dev)
    r = ddist.all_reduce_sum(t.clone())
    res["all_reduce_sum"] = bool(r.is_cuda and torch.equal(r, t))
    a = ddist.all_gather_rows(t.reshape(30, 40))
    res["all_gather_rows"] = bool(a.is_cuda and a.shape == (1, 30, 40) and torch.equal(a[0], t.reshape(30, 40)))
    os.environ["DGDM_TORCH_SEED"] = "4242"
    seed = ddist.sync_start_stream_seed()
    res["seed"] = [int(seed), int(torch.initial_seed())]
    os.environ.pop("DGDM_TORCH_SEED")
    seed2 = ddist.sync_start_stream_seed()                       # rank 0's own seed, broadcast over RCCL
    res["seed_broadcast"] = [int(seed2), int(torch.initial_seed())]
    # the barrier + max-over-ranks timing of bench.py
    tm = torch.tensor([1.5], device=dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dist.barrier()
    res["max"] = float(tm.item())
    torch.cuda.synchronize()
    res["rccl_mapped"] = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "rccl" in ln})[:2]
    dist.destroy_process_group()
    print("RCCL_WORKER " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
