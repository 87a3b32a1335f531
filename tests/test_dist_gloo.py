"""world_size-2 CPU test (gloo) of the pair sharding + final gather used on N > 1 GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dgdm_amd.dist import gather_pairs, run_sharded, shard_range


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(n, r, world)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs = [(i % 3, f"obj{i}") for i in range(n_pairs)]

    def run_local(mine):          # stands in for sampler.guided_chains: row i carries its global pair index
        idx = [pairs.index(p) for p in mine]
        return torch.tensor(idx, dtype=torch.float32).reshape(-1, 1, 1, 1).expand(-1, 2, 3, 1).contiguous()

    out = run_sharded(pairs, run_local)
    ok = out.shape == (n_pairs, 2, 3, 1) and torch.equal(out[:, 0, 0, 0], torch.arange(n_pairs, dtype=torch.float32))
    again = gather_pairs(run_local([pairs[i] for i in shard_range(n_pairs, rank, world)]), n_pairs)
    q.put((rank, bool(ok and torch.equal(again, out))))
    dist.destroy_process_group()


def test_two_ranks_gather_in_pair_order():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    for n_pairs in (5, 8):
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
        assert res == [(0, True), (1, True)]
        port += 1
