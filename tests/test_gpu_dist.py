"""Multi-GPU plumbing on the one-GPU box: RCCL itself (a process group of one rank on the "nccl" backend) and the sharding of
BASELINE configs[3] (256 (object x objective) pairs over 8 ranks) replayed rank block by rank block in one process."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from dgdm_amd import dist as ddist
from dgdm_amd import engine, sampler, synth
from dgdm_amd.scheduler import DDIMScheduler
from tests import util
from tests.test_gpu_parity import dev      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_group_of_one_rank():
    """librccl loads and the device branches of dgdm_amd/dist.py (gather_pairs, all_reduce_sum, all_gather_rows, the seed broadcast)
    run on device tensors through the "nccl" backend.  Own process: a process group cannot be torn down and rebuilt inside pytest."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL_WORKER ")][-1]
    r = json.loads(line[len("RCCL_WORKER "):])
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["gather_pairs"] and r["all_reduce_sum"] and r["all_gather_rows"]
    assert r["seed"] == [4242, 4242] and r["seed_broadcast"][0] == r["seed_broadcast"][1] and r["max"] == 1.5
    assert r["rccl_mapped"], "the worker process did not map librccl"         # the library the collectives went through


def test_config3_rank_blocks_equal_one_batch(dev):      # noqa: F811
    """BASELINE configs[3]: 256 (object x objective) pairs x B = 32 fingers at the shipped 3-D grid, block-partitioned over 8 ranks.
    Ranks 0 and 7 are replayed here one after the other - their 32-pair blocks, their objects, and their FPS start draws taken from the
    ONE generator stream the reference's sequential loops would consume (every rank walks it over all 256 chains and skips the others',
    sampler.draw_chain_starts keep=) - and compared bit for bit with the same 64 pairs run as a single batch."""
    B, G, P, L, N, sub, T, S = 32, 45, 5, 42, 512, 512, 15, 5
    n_pairs, world = 256, 8
    objectives = [o for o in synth.OBJECTIVES_12 if o != 'convergence']
    chains = [(i, objectives[i % len(objectives)]) for i in range(n_pairs)]
    net = engine.Unet1d(util.unet_sd(11))
    dyn = engine.Dynamics(3, util.dyn3d_sd(33), L)
    s = DDIMScheduler(num_train_timesteps=T)
    s.set_timesteps(S)
    noise = synth.synth_noise(0, B, L).to(dev)
    spec = ddist.GuidanceSpec(dyn, B, G, P, (-1.0, 1.0), T, N, sub)
    outs, pres, blocks = {}, {}, {}
    for rank in (0, 7):
        mine, local_objects, local_chains = ddist.shard_chains(chains, rank, world)
        assert list(mine) == list(range(32 * rank, 32 * rank + 32)) and local_objects == list(mine)
        torch.manual_seed(20260)                              # every rank starts the walk from the same generator state (dist.sync_start_stream_seed)
        sweep, step = sampler.draw_chain_starts(spec, chains, S, keep=mine)
        assert step.shape == (S, 32, 2 * spec.rows)
        # the rank's first chain sits behind the draws of all earlier chains in the stream
        r = sampler.TorchRng(seed=20260)
        r.randint(512, mine.start * S * 2 * spec.rows, skip=True)
        assert np.array_equal(step[:, 0], r.randint(512, S * 2 * spec.rows).reshape(S, -1))
        objs = torch.stack([synth.synth_object_3d(1000 + i, N) for i in mine]).to(dev)
        guid = spec.build(objs, 32)
        outs[rank] = sampler.guided_chains(net, guid, s, 'point_3d', noise, local_chains, predrawn=(sweep, step))
        pres[rank], blocks[rank] = step, (objs, local_chains)
        del guid
        torch.cuda.empty_cache()
    objs = torch.cat([blocks[0][0], blocks[7][0]])
    both = [(i, o) for i, (_, o) in enumerate(blocks[0][1] + blocks[7][1])]
    guid = spec.build(objs, 64)
    step = np.ascontiguousarray(np.concatenate([pres[0], pres[7]], axis=1))
    out = sampler.guided_chains(net, guid, s, 'point_3d', noise, both, predrawn=([None] * 64, step))
    assert out.shape == (64, B, L, 1) and bool(torch.isfinite(out).all())
    assert torch.equal(out[:32], outs[0]) and torch.equal(out[32:], outs[7])
    # and the gather puts the blocks back in pair order (single process: identity; the 2-rank version is tests/test_dist_gloo.py)
    assert torch.equal(ddist.gather_pairs(outs[0], 32), outs[0])


def test_bench_rank_body_on_rccl_world_size_1():
    """The WHOLE rank body of bench.py on RCCL at world size 1 (`--force-group`: nccl process group, the barriers around the timed region, the
    gather of the final samples, the max-over-ranks reduction, the JSON line) against the plain single-process run: the same value
    within 5 % over 10 steps (measured: 2-3 % lower - not the gather, which runs asynchronously on RCCL's stream, but the table build,
    12.6 instead of 11.9 ms per step: RCCL's stream takes one of the hardware queues the build's streams share; box-to-box noise of two
    back-to-back runs is 1-2 %).  What an 8-GPU node adds to this is the transport."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    vals = {}
    for tag, extra in (("plain", []), ("rccl", ["--force-group"])):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-extra", "--no-2d"] + extra, cwd=root, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        last = r.stdout.strip().splitlines()[-1]                      # what a tail-capturing driver keeps
        assert last.startswith("{") and len(last) < 4096, (len(last), r.stdout[-600:])
        line = json.loads(last)
        assert line["n_gpus"] == 1 and line["steps"] == 10 and line["roofline"]["frac"] > 0 and line["roofline"]["avg_launch_ms"] > 0
        assert all(not isinstance(v, (dict, list)) for v in line["roofline"].values())
        vals[tag] = line["value"]
    assert abs(vals["rccl"] / vals["plain"] - 1) < 0.05, vals
