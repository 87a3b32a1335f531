"""world_size-2 CPU test (gloo) of the pair sharding + final gather used on N > 1 GPUs."""
import os
import socket

import numpy as np
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
    # the asynchronous form (bench.py: a step's gather beside the next step's kernels): the handle's wait() gives the same tensor
    later = gather_pairs(run_local([pairs[i] for i in shard_range(n_pairs, rank, world)]), n_pairs, async_op=True)
    q.put((rank, bool(ok and torch.equal(again, out) and torch.equal(later.wait(), out) and torch.equal(later.wait(), out))))
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


# ------------------------------------------------------------------------------------------------ rank-count-invariant start streams
class _Spec:
    """Geometry of a 3-D guidance handle (what sampler.draw_chain_starts reads), without a GPU."""

    def __init__(self, B=2, G=3, P=2, N=512, sub=5):
        from dgdm_amd.dist import GuidanceSpec
        self.spec = GuidanceSpec(None, B, G, P, (-1.0, 1.0), 15, N, sub)


CHAINS = [(0, 'rotate'), (1, 'convergence'), (2, 'shift_up'), (0, 'convergence'), (1, 'clockwise_left'), (2, 'rotate'), (0, 'shift_down')]


def _fake_guided_chains(unet, guid, sched, mode, noise, chains, unguided=None, starts=None, trace=None, predrawn=None):
    """Stand-in for sampler.guided_chains on a box without a GPU: every chain's 'sample' is a checksum of exactly the inputs the
    real loop would consume - its object (through the local bank), its objective and its FPS start draws."""
    sweep, step = predrawn
    B, L, _ = noise.shape
    out = torch.zeros((len(chains), B, L, 1))
    for k, (oi, o) in enumerate(chains):
        out[k, 0, 0, 0] = float(guid.bank[oi].sum())
        out[k, 0, 1, 0] = float(sum(map(ord, o)))
        out[k, 0, 2, 0] = float(step[:, k].astype(np.float64).sum() % 65521)
        out[k, 0, 3, 0] = float(step[-1, k, -7:].sum())
        out[k, 1, 0, 0] = -1.0 if sweep[k] is None else float(sweep[k].astype(np.float64).sum() % 65521)
    return out


class _FakeGuid:
    def __init__(self, bank):
        self.bank = bank


def _stream_worker(rank, world, port, q):
    import numpy as np      # noqa: F811
    from dgdm_amd import dist as dd
    from dgdm_amd import sampler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = _Spec().spec
    S = 5
    sched = type("S", (), {"timesteps": list(range(S))})()
    objects = torch.arange(3 * 512 * 3, dtype=torch.float32).reshape(3, 512, 3)
    noise = torch.zeros(2, 42, 1)
    ok = True
    # (1) the draws a rank keeps are the single-process draws of its chains, and the global generator ends in the same state
    torch.manual_seed(7)
    sweep_all, step_all = sampler.draw_chain_starts(spec, CHAINS, S)
    state_all = torch.get_rng_state()
    mine = dd.shard_range(len(CHAINS), rank, world)
    torch.manual_seed(7)
    sweep, step = sampler.draw_chain_starts(spec, CHAINS, S, keep=mine)
    ok = ok and torch.equal(torch.get_rng_state(), state_all) and np.array_equal(step, step_all[:, mine.start:mine.stop])
    for k, c in enumerate(mine):
        ok = ok and ((sweep[k] is None and sweep_all[c] is None) or np.array_equal(sweep[k], sweep_all[c]))
    # (2) the sharded loop hands every chain those draws and gathers the samples in the original pair order
    sampler.guided_chains, real = _fake_guided_chains, sampler.guided_chains
    try:
        torch.manual_seed(7)
        got = dd.guided_chains_sharded(None, spec, sched, 'point_3d', noise, objects, CHAINS, build=lambda o, n: _FakeGuid(o))
        want = _fake_guided_chains(None, _FakeGuid(objects), sched, 'point_3d', noise, CHAINS, predrawn=(sweep_all, step_all))
        ok = ok and torch.equal(got, want) and torch.equal(torch.get_rng_state(), state_all)
        # (3) per-pair streams: a pair's draws depend on its global index only
        streams = [sampler.pair_stream(512, 5, 99, 1000 + c) for c in range(len(CHAINS))]
        _, one = sampler.draw_chain_starts(spec, CHAINS, S, streams=streams)
        streams = [sampler.pair_stream(512, 5, 99, 1000 + c) for c in range(len(CHAINS))]
        _, part = sampler.draw_chain_starts(spec, CHAINS, S, keep=mine, streams=streams)
        ok = ok and np.array_equal(part, one[:, mine.start:mine.stop])
    finally:
        sampler.guided_chains = real
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_start_streams_match_single_process():
    """FPS starts and gathered pair order of a 2-rank run equal the single-process ones."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_multi_object_stream_split():
    """guided_multi_object_sharded's draw walk: per step, object after object; a rank keeps its objects' draws and skips the rest."""
    from dgdm_amd import sampler
    spec = _Spec().spec
    n_obj, S = 3, 4
    torch.manual_seed(3)
    st = sampler.StartStream(512, 5)
    full = [[st.call(spec.rows) for _ in range(n_obj)] for _ in range(S)]
    end = torch.get_rng_state()
    for world in (2, 3):
        for rank in range(world):
            from dgdm_amd.dist import shard_range
            mine = shard_range(n_obj, rank, world)
            torch.manual_seed(3)
            st = sampler.StartStream(512, 5)
            for si in range(S):
                per = [st.call(spec.rows) if j in mine else st.skip(spec.rows) for j in range(n_obj)]
                for j in mine:
                    assert np.array_equal(per[j], full[si][j])
            assert torch.equal(torch.get_rng_state(), end)


# ------------------------------------------------------------------------------------------------ data-parallel training collectives
def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dgdm_amd.dist import all_gather_rows, all_reduce_sum
    g = all_reduce_sum(torch.arange(6, dtype=torch.float32) * (rank + 1))            # the gradient buffers of Trainer's data-parallel step
    rows = all_gather_rows(torch.full((3, 2), float(rank)))
    ok = torch.equal(g, torch.arange(6, dtype=torch.float32) * 3) and rows.shape == (2, 3, 2) and torch.equal(rows[:, 0, 0], torch.tensor([0.0, 1.0]))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_data_parallel_collectives():
    """all_reduce_sum / all_gather_rows as Trainer._run uses them (summed gradients, gathered prediction chunks + loss shares)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
