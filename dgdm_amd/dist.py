"""Multi-GPU execution of the guided path: one process per GPU, independent (object x objective) pairs
block-partitioned over the ranks, ONE collective at the end.

This replaces the reference's only multi-GPU mechanism on the path, ``nn.DataParallel`` around the dynamics model
(generator/train.py:86,88: per-call scatter / replicate / gather inside every cond_fn).  Chains never exchange data,
every rank holds a full weight replica (~20 MB), so nothing is communicated inside the denoise loop; the final
samples [pairs, B, L, 1] are all-gathered over RCCL (backend "nccl" on ROCm) - 1.4 MB for 256 pairs, latency-bound.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block partition; the first `n_items % world` ranks take one extra item."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return range(lo, lo + q + (1 if rank < r else 0))


class _Gathered:
    """An all_gather in flight (gather_pairs(..., async_op=True)): wait() returns the gathered tensor.  Until then the collective
    runs on the backend's own stream behind what produced `local`, beside whatever the caller enqueues next."""

    def __init__(self, work, finish):
        self._work, self._finish = work, finish

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._finish()


def gather_pairs(local: torch.Tensor, n_items: int, group=None, async_op: bool = False):
    """all_gather of per-rank results [n_local, ...] (block partition of `n_items`) -> [n_items, ...] on every rank.
    async_op=True: returns a handle whose wait() gives the tensor - the gather then overlaps the caller's next kernels (RCCL runs it on
    its own stream behind the producer of `local`; bench.py gathers a step's samples while the next step's tables are being built)."""
    if not dist.is_available() or not dist.is_initialized():
        return _Gathered(None, lambda: local) if async_op else local
    # an initialised group of ONE rank still goes through the collective (RCCL on the device tensor): that is how the 1-GPU box
    # exercises this branch (tests/test_gpu_dist.py)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    most = max(len(shard_range(n_items, r, world)) for r in range(world))
    # RCCL ("nccl") gathers device tensors in place; under gloo (CPU tests, or ranks sharing a GPU) the payload goes through host memory
    dev = local.device
    via_host = local.is_cuda and dist.get_backend(group) != "nccl"
    src = local.cpu() if via_host else local
    pad = torch.zeros((most,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[:src.shape[0]] = src
    bufs = [torch.empty_like(pad) for _ in range(world)]

    def finish():
        out = torch.cat([bufs[r][:len(shard_range(n_items, r, world))] for r in range(world)], dim=0)
        return out.to(dev) if via_host else out
    if async_op and not via_host:
        return _Gathered(dist.all_gather(bufs, pad, group=group, async_op=True), finish)
    dist.all_gather(bufs, pad, group=group)
    return _Gathered(None, finish) if async_op else finish()


def run_sharded(pairs: Sequence[Tuple[int, str]], run_local: Callable[[List[Tuple[int, str]]], torch.Tensor], group=None) -> torch.Tensor:
    """Runs `run_local` on this rank's block of `pairs` and returns all pairs' samples in the original order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = [pairs[i] for i in shard_range(len(pairs), rank, world)]
    return gather_pairs(run_local(mine), len(pairs), group)


def world_rank(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Joins the process group a launcher described in the environment (torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*);
    a no-op for a single process.  backend: "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo".  Returns (world, rank, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend is None:       # DGDM_DIST_BACKEND=gloo: functional runs on a box with fewer GPUs than ranks (ranks then share the GPUs)
            backend = os.environ.get("DGDM_DIST_BACKEND") or ("nccl" if torch.cuda.device_count() > 0 else "gloo")
        if torch.cuda.device_count() > 0:
            local = local % torch.cuda.device_count() if backend != "nccl" else local
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    sync_start_stream_seed()
    return world, rank, local


def _src0(group=None) -> int:
    """The GLOBAL rank of the group's rank 0: what dist.broadcast's `src` means (it is 0 only for the default group)."""
    return dist.get_global_rank(group, 0) if group is not None else 0


def sync_start_stream_seed(group=None) -> int:
    """One seed for the torch CPU generator on every rank, so that all ranks walk the same FPS start stream
    (``guided_chains_sharded``).  The reference never seeds that generator (its 3-D results differ from run to run);
    DGDM_TORCH_SEED pins it - for one process as well - otherwise rank 0's own (random) initial seed is broadcast."""
    env = os.environ.get("DGDM_TORCH_SEED")
    seed = int(env) if env else int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
    grouped = dist.is_available() and dist.is_initialized()
    if grouped:
        t = torch.tensor([seed], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=_src0(group), group=group)
        seed = int(t.item())
    if not env and not (grouped and dist.get_world_size(group) > 1):
        # one process (or a group of one rank), nothing pinned: leave the generator as it is - re-seeding would rewind a generator
        # that has already been drawn from
        return seed
    torch.manual_seed(seed)
    return seed


def shard_chains(chains: Sequence[Tuple[int, str]], rank: int, world: int):
    """This rank's block of `chains` [(object index, opt_obj)]: (global index range, the object indices it needs in first-use
    order, its chains re-indexed into that local object bank)."""
    mine = shard_range(len(chains), rank, world)
    local_objects: List[int] = []
    local_chains: List[Tuple[int, str]] = []
    for i in mine:
        oi, o = chains[i]
        if oi not in local_objects:
            local_objects.append(oi)
        local_chains.append((local_objects.index(oi), o))
    return mine, local_objects, local_chains


class GuidanceSpec:
    """What defines a guidance handle apart from its objects.  Quacks like ``engine.Guidance`` for the start-stream arithmetic
    (rows per cond_fn call, rows per centre sweep), so a rank can walk the draw stream without building a handle."""

    def __init__(self, dyn, batch: int, grid_size: int, num_pos: int, ori_range, num_train_timesteps: int, num_object_points: int,
                 sub_batch_size: int = 0, contraction_dtype: str = "f32"):
        self.dyn, self.batch, self.grid_size, self.num_pos, self.ori_range = dyn, batch, grid_size, num_pos, tuple(ori_range)
        self.num_train_timesteps, self.num_object_points, self.sub_batch_size = num_train_timesteps, num_object_points, sub_batch_size
        self.contraction_dtype = contraction_dtype
        self.rows = batch * grid_size * num_pos * num_pos
        self.sweep_rows = batch * grid_size
        self.starts_per_call = 2 * self.rows
        self.cfg = self

    def build(self, objects: torch.Tensor, n_chains: int):
        from . import engine
        g = engine.Guidance(self.dyn, self.batch, self.grid_size, self.num_pos, self.ori_range, max(1, n_chains), self.num_train_timesteps,
                            self.num_object_points, self.sub_batch_size, max_objects=max(1, objects.shape[0]),
                            contraction_dtype=self.contraction_dtype)
        g.set_objects(objects)
        return g


def guided_chains_sharded(unet, spec: GuidanceSpec, sched, mode: str, noise: torch.Tensor, objects: torch.Tensor,
                          chains: Sequence[Tuple[int, str]], unguided: Optional[torch.Tensor] = None, starts=None,
                          streams=None, group=None, build: Optional[Callable] = None) -> torch.Tensor:
    """``sampler.guided_chains`` for the (object x objective) pairs `chains` over the object bank `objects`, block-partitioned
    over the ranks of `group`; returns all chains' samples [n_chains, B, L, 1] in the original order on every rank.

    Each rank builds the tables of the objects ITS chains use (``spec.build`` or `build(local_objects, n_local_chains)`) and
    runs its chains; nothing is communicated inside the denoise loop and the final samples are all-gathered once (SURVEY.md
    §8(e)).  3-D: the FPS start draws are rank-count-invariant - every rank walks the reference's single generator stream over
    ALL chains (host only: `starts` or the global CPU generator) and keeps its own block, unless per-chain `streams` are given."""
    from . import sampler
    world, rank = world_rank(group)
    mine, local_objects, local_chains = shard_chains(chains, rank, world)
    B, L, _ = noise.shape
    S = len(sched.timesteps)
    predrawn = None
    if mode == 'point_3d':
        predrawn = sampler.draw_chain_starts(spec, chains, S, starts, keep=mine, streams=streams)
    if len(local_chains):
        guid = (build or spec.build)(objects[local_objects].to(noise.device), len(local_chains))
        local = sampler.guided_chains(unet, guid, sched, mode, noise, local_chains, unguided=unguided, predrawn=predrawn)
    else:
        local = torch.zeros((0, B, L, 1), dtype=torch.float32, device=noise.device)
    return gather_pairs(local, len(chains), group)


def gather_rows(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """Alias of gather_pairs for per-step payloads (the stacked gradients of the multi-object loop)."""
    return gather_pairs(local, n_items, group)


def guided_multi_object_sharded(unet, spec: GuidanceSpec, sched, mode: str, noise: torch.Tensor, objects: torch.Tensor, opt_obj: str,
                                starts=None, group=None, build: Optional[Callable] = None, on_step=None) -> torch.Tensor:
    """``sampler.guided_multi_object`` (generator/diffusion.py:637-647: ONE chain whose step uses the mean of n_obj cond_fn
    gradients) with the objects block-partitioned over the ranks.  Unlike the per-object chains this loop has a real exchange
    step: every denoise step the ranks all-gather their objects' gradients [n_obj, B, L] (a few KB over RCCL) and then all take
    the same scheduler step on the full stack, in object order - so every rank holds the single-process result bit for bit.
    3-D: per step the reference draws the FPS starts object after object (:641-643); every rank walks that stream and keeps its objects'."""
    from . import engine, sampler
    world, rank = world_rank(group)
    n_obj, (B, L, _) = objects.shape[0], noise.shape
    if opt_obj == 'convergence':
        raise ValueError("the reference never runs the multi-object loop with 'convergence' (generator/diffusion.py:337)")
    mine = shard_range(n_obj, rank, world)
    is3d = mode == 'point_3d'
    if is3d:
        starts = starts or sampler.StartStream(spec.num_object_points, spec.sub_batch_size)
    guid = (build or spec.build)(objects[mine.start:mine.stop].to(noise.device), len(mine)) if len(mine) else None
    objectives = [engine.make_objective(opt_obj, k) for k in range(len(mine))]
    scale = sampler.classifier_scale(mode, opt_obj, multi=True)
    x = noise.reshape(B, L).contiguous().to(torch.float32)
    for i, t in enumerate(sched.timesteps):
        t = int(t)
        eps = unet.forward(x.reshape(B, L, 1), torch.full((B,), t, dtype=torch.int32, device=x.device)).reshape(B, L)
        st = None
        if is3d:
            per_obj = [starts.call(spec.rows) if j in mine else starts.skip(spec.rows) for j in range(n_obj)]
            st = np.concatenate([per_obj[j] for j in mine]) if len(mine) else None
        if guid is not None:
            g = guid.grad(x.reshape(1, B, L).expand(len(mine), -1, -1).contiguous(), t, objectives, None, st)
        else:
            g = torch.zeros((0, B, L), dtype=torch.float32, device=x.device)
        g = gather_rows(g, n_obj, group)
        x = engine.ddim_guided_step(x, eps, g, n_obj, sched.coefficients(t), scale)
        if on_step is not None:
            on_step(i, x.reshape(B, L, 1))
    return x.reshape(B, L, 1)


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum over the ranks (gradients of data-parallel training): RCCL on the device tensor; through host memory under gloo."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """[n, ...] of equal shape on every rank -> [world, n, ...] on every rank."""
    world, _ = world_rank(group)
    if not (dist.is_available() and dist.is_initialized()):
        return t[None]
    via_host = t.is_cuda and dist.get_backend(group) != "nccl"
    src = t.cpu() if via_host else t.contiguous()
    bufs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(bufs, src, group=group)
    out = torch.stack(bufs)
    return out.to(t.device) if via_host else out


def broadcast_from_rank0(t: torch.Tensor, group=None) -> torch.Tensor:
    """Rank 0's tensor on every rank (BatchNorm running statistics of data-parallel training): RCCL on the device tensor; through host
    memory under gloo."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.broadcast(h, src=_src0(group), group=group)
        return h.to(t.device)
    dist.broadcast(t, src=_src0(group), group=group)
    return t
