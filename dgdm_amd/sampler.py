"""Denoise loops of ``generator/diffusion.py`` (:193-201, :249-256, :570-576, :637-647) on the HIP path.

Independent chains - the (object, objective) pairs ``Diffusion.guided_sample`` walks through one after
the other (:561) - are advanced together as an extra batch axis: one U-Net launch, one guidance
launch sequence and one scheduler launch per denoise step for all of them.  The chains do not
interact, so the results equal the reference's sequential evaluation; what has to be preserved is
the order in which the reference consumes the torch CPU generator for the FPS starts
(dynamics/models/pointnet2_utils.py:83), which ``StartStream`` replays.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import engine
from .engine import Guidance, Unet1d, make_objective
from .scheduler import DDIMScheduler

SCALE_2D, SCALE_2D_CONV, SCALE_3D, SCALE_3D_CONV = 0.001, 10.0, 0.5, 0.8      # generator/diffusion.py:30-33


def classifier_scale(mode: str, opt_obj: str, multi: bool = False) -> float:
    """generator/diffusion.py:549-560 (per-object loop) and :631-636 (multi-object loop)."""
    if mode == 'point':
        return SCALE_2D_CONV if (opt_obj == 'convergence' and not multi) else SCALE_2D
    if mode == 'point_3d':
        return SCALE_3D_CONV if (opt_obj == 'convergence' and not multi) else SCALE_3D
    return 0.001


class StartStream:
    """FPS start indices in the reference's draw order.

    One classifier call on ``rows`` rows with sub-batch size ``sub`` draws, per sub-batch, ``torch.randint(0, N, (n,))``
    for sa1 and ``torch.randint(0, 512, (n,))`` for sa2 from the global CPU generator
    (pointnet2_utils.py:83 reached via generator/diffusion.py:495-498 and :524-526).

    ``generator``: a private ``torch.Generator`` instead of the global one (per-pair streams, ``pair_stream``)."""

    def __init__(self, num_points: int, sub_batch_size: int, forced: Optional[Sequence[torch.Tensor]] = None,
                 generator: Optional[torch.Generator] = None):
        self.N, self.sub = int(num_points), int(sub_batch_size)
        self._forced = list(forced) if forced is not None else None
        self._gen = generator

    def _randint(self, high: int, n: int) -> torch.Tensor:
        if self._gen is not None:
            return torch.randint(0, high, (n,), dtype=torch.long, generator=self._gen)
        return torch.randint(0, high, (n,), dtype=torch.long)

    def _draw(self, high: int, n: int) -> np.ndarray:
        if self._forced is not None:
            s = self._forced.pop(0)
            assert s.shape == (n,), (s.shape, n)
            return s.numpy().astype(np.int64)
        return self._randint(high, n).numpy()

    def call(self, rows: int) -> np.ndarray:
        """The 2*rows indices one classifier call over `rows` rows consumes, in draw order:
        [sub-batch 0: sa1 x n0, sa2 x n0 | sub-batch 1: ...] - the layout dgdm_dyn3d_guidance_grad takes."""
        if self._forced is None and self.N == 512:
            # same range for both layers: consecutive randint calls on the CPU generator concatenate
            # (tests/test_host_logic.py::test_start_stream_matches_reference_draws), so one call does it
            return self._randint(512, 2 * rows).numpy()
        out = np.empty(2 * rows, dtype=np.int64)
        for r0 in range(0, rows, self.sub):
            n = min(self.sub, rows - r0)
            out[2 * r0:2 * r0 + n] = self._draw(self.N, n)
            out[2 * r0 + n:2 * r0 + 2 * n] = self._draw(512, n)
        return out

    def skip(self, rows: int) -> None:
        """Consume the draws of one classifier call without keeping them (a rank replaying the global stream past chains
        that belong to other ranks, dgdm_amd/dist.py)."""
        self.call(rows)


def pair_stream(num_points: int, sub_batch_size: int, seed: int, pair_index: int) -> StartStream:
    """A start stream of its own for pair `pair_index` (seed-derived), for workloads that have no reference draw order to keep
    (bench.py's synthetic pair batches): the draws of a pair depend on (seed, pair index) only, not on the rank that runs it or on
    how many ranks there are."""
    g = torch.Generator()
    g.manual_seed((int(seed) * 1_000_003 + int(pair_index)) & 0x7FFFFFFFFFFFFFFF)
    return StartStream(num_points, sub_batch_size, generator=g)


def unguided_sample(unet: Unet1d, sched: DDIMScheduler, x: torch.Tensor, on_step=None) -> torch.Tensor:
    """S x [eps-net ; DDIM step]  (generator/diffusion.py:193-201, :249-256).  on_step(i, x_i, eps_i): the harness's per-step hook
    (plots, noise-prediction loss); it forces a device->host copy per step, as the reference's own plotting does."""
    B = x.shape[0]
    x = x.clone()
    for i, t in enumerate(sched.timesteps):
        ts = torch.full((B,), int(t), dtype=torch.int32, device=x.device)
        eps = unet.forward(x, ts)
        x = engine.ddim_guided_step(x, eps, None, 0, sched.coefficients(int(t)), 0.0)
        if on_step is not None:
            on_step(i, x, eps)
    return x


def convergence_centers(guid: Guidance, mode: str, unguided: torch.Tensor, objects_of_chain: Sequence[int],
                        starts: Optional[np.ndarray] = None) -> torch.Tensor:
    """``Diffusion.get_convergence_centers`` (:506-539) for several objects at once -> (n_chains, B) int64."""
    from .dynamics import metrics
    nc, B = len(objects_of_chain), unguided.shape[0]
    x = unguided.reshape(1, B, -1).expand(nc, -1, -1).contiguous()
    logits = guid.sweep(x, objects_of_chain, starts).cpu()                       # (nc, B*G, 3), row = g*B + b
    if mode == 'point_3d':
        thr = torch.tensor(0.02) / torch.tensor(0.0312)                          # threshold/std (:116-118)
    else:
        thr = torch.tensor(0.03) / torch.tensor(0.0565)
    G = guid.cfg.grid_size
    out = torch.zeros((nc, B), dtype=torch.int64)
    for c in range(nc):
        d0 = logits[c, :, 0]
        prof = torch.where(d0 > thr, 2.0, torch.where(d0 < -thr, 0.0, 1.0))      # :532
        for i in range(B):
            lengths, centers = metrics.convergence_mode_three_class(prof[torch.arange(i, B * G, B)])
            out[c, i] = centers[torch.argmax(lengths)]
    return out


def draw_chain_starts(guid: Guidance, chains: Sequence[Tuple[int, str]], n_steps: int, starts: Optional[StartStream] = None,
                      keep: Optional[range] = None, streams: Optional[Sequence[StartStream]] = None):
    """FPS starts of a batch of 3-D chains in the order the reference's sequential loops consume the generator:
    chain after chain (generator/diffusion.py:561); inside a chain the centre sweep (:563) and then every step's cond_fn (:574).

    keep: only the chains of this index range are returned (sweep list / step array of len(keep) chains); the draws of the
    others are consumed and dropped, so that a rank holding a block of the chains sees exactly the numbers a single process
    would have handed those chains (dgdm_amd/dist.py).  streams: one StartStream per chain instead of the shared one (then
    nothing has to be skipped: chains outside `keep` are not touched)."""
    nc = len(chains)
    keep = range(nc) if keep is None else keep
    if streams is None:
        starts = starts or StartStream(guid.cfg.num_object_points, guid.cfg.sub_batch_size)
    sweep: List[Optional[np.ndarray]] = [None] * len(keep)
    step = np.zeros((n_steps, len(keep), guid.starts_per_call), dtype=np.int64)
    for c, (_, o) in enumerate(chains):
        mine = c in keep
        if streams is not None:
            if not mine:
                continue
            st = streams[c]
        else:
            st = starts
        k = c - keep.start
        if o == 'convergence':
            if mine:
                sweep[k] = st.call(guid.sweep_rows)
            else:
                st.skip(guid.sweep_rows)
        for si in range(n_steps):
            if mine:
                step[si, k] = st.call(guid.rows)
            else:
                st.skip(guid.rows)
    return sweep, step


def guided_chains(unet: Unet1d, guid: Guidance, sched: DDIMScheduler, mode: str, noise: torch.Tensor,
                  chains: Sequence[Tuple[int, str]], unguided: Optional[torch.Tensor] = None,
                  starts: Optional[StartStream] = None, trace: Optional[list] = None, predrawn=None) -> torch.Tensor:
    """``Diffusion.guided_sample`` loop bodies (:561-576) for the chains [(object index, opt_obj), ...].

    noise (B, L, 1) is shared by all chains (:570).  Returns (n_chains, B, L, 1)."""
    nc, (B, L, _) = len(chains), noise.shape
    dev = noise.device
    is3d = mode == 'point_3d'
    S = len(sched.timesteps)
    objectives = [make_objective(o, oi) for oi, o in chains]
    # --- replay of the reference's RNG consumption: chain after chain; inside a chain the centre sweep, then the steps
    sweep_starts: List[Optional[np.ndarray]] = [None] * nc
    step_starts = None
    if is3d:
        sweep_starts, step_starts = predrawn if predrawn is not None else draw_chain_starts(guid, chains, S, starts)
    # --- 'convergence' chains: centres from the unguided sample, then row coefficients
    rowcoef = None
    conv = [c for c, (_, o) in enumerate(chains) if o == 'convergence']
    if conv:
        assert unguided is not None, "opt_obj='convergence' needs the unguided sample (generator/diffusion.py:563)"
        st = np.concatenate([sweep_starts[c] for c in conv]) if is3d else None
        centers = convergence_centers(guid, mode, unguided, [chains[c][0] for c in conv], st)
        rc = np.zeros((nc, guid.rows), dtype=np.float32)
        for k, c in enumerate(conv):
            rc[c] = guid.rowcoef(centers[k])
        rowcoef = torch.from_numpy(rc).to(dev)
    scales = [classifier_scale(mode, o) for _, o in chains]
    if trace is None:         # the loop itself runs inside the library (one call); the Python loop below is kept for traced runs
        out = engine.guided_chains_run(unet, guid, noise.reshape(B, L), nc, 1, objectives, rowcoef,
                                       np.ascontiguousarray(step_starts) if is3d else None, [int(t) for t in sched.timesteps],
                                       [sched.coefficients(int(t)) for t in sched.timesteps], scales)
        return out.reshape(nc, B, L, 1)
    x = noise.reshape(1, B, L).expand(nc, -1, -1).contiguous().to(torch.float32)
    for si, t in enumerate(sched.timesteps):
        t = int(t)
        if si == 0 and nc > 1:
            # every chain starts from the same noise (generator/diffusion.py:570: `sample = noise.clone()` per object), so the first
            # eps-net call has nc identical copies of the B fingers as its batch: evaluate it once (same input, same kernel, same bits)
            ts = torch.full((B,), t, dtype=torch.int32, device=dev)
            eps = unet.forward(x[0].reshape(B, L, 1), ts).reshape(1, B, L).expand(nc, -1, -1).contiguous()
        else:
            ts = torch.full((nc * B,), t, dtype=torch.int32, device=dev)
            eps = unet.forward(x.reshape(nc * B, L, 1), ts).reshape(nc, B, L)
        g = guid.grad(x, t, objectives, rowcoef, step_starts[si].reshape(-1) if is3d else None)
        if trace is not None:
            trace.append((eps.clone(), g.clone()))
        coef = sched.coefficients(t)
        if len(set(scales)) == 1:
            x = engine.ddim_guided_step(x, eps, g, 1, coef, scales[0])
        else:
            x = torch.stack([engine.ddim_guided_step(x[c], eps[c], g[c], 1, coef, scales[c]) for c in range(nc)])
    return x.reshape(nc, B, L, 1)


def guided_multi_object(unet: Unet1d, guid: Guidance, sched: DDIMScheduler, mode: str, noise: torch.Tensor,
                        object_indices: Sequence[int], opt_obj: str, starts: Optional[StartStream] = None, on_step=None) -> torch.Tensor:
    """``Diffusion.guided_sample_multi_object`` loop (:637-647): one chain, gradient = mean over the objects."""
    n_obj, (B, L, _) = len(object_indices), noise.shape
    dev = noise.device
    is3d = mode == 'point_3d'
    objectives = [make_objective(opt_obj, oi) for oi in object_indices]
    if opt_obj == 'convergence':
        raise ValueError("the reference never runs the multi-object loop with 'convergence' (generator/diffusion.py:337)")
    if is3d:
        starts = starts or StartStream(guid.cfg.num_object_points, guid.cfg.sub_batch_size)
    scale = classifier_scale(mode, opt_obj, multi=True)
    if on_step is None:       # one library call for the whole loop; per step the reference draws object after object (:641-643)
        S = len(sched.timesteps)
        st = np.stack([np.concatenate([starts.call(guid.rows) for _ in object_indices]) for _ in range(S)]) if is3d else None
        out = engine.guided_chains_run(unet, guid, noise.reshape(B, L), 1, n_obj, objectives, None, st, [int(t) for t in sched.timesteps],
                                       [sched.coefficients(int(t)) for t in sched.timesteps], [scale])
        return out.reshape(B, L, 1)
    x = noise.reshape(B, L).contiguous().to(torch.float32)
    for i, t in enumerate(sched.timesteps):
        t = int(t)
        ts = torch.full((B,), t, dtype=torch.int32, device=dev)
        eps = unet.forward(x.reshape(B, L, 1), ts).reshape(B, L)
        st = np.concatenate([starts.call(guid.rows) for _ in object_indices]) if is3d else None      # object after object (:641-643)
        g = guid.grad(x.reshape(1, B, L).expand(n_obj, -1, -1).contiguous(), t, objectives, None, st)
        x = engine.ddim_guided_step(x, eps, g, n_obj, sched.coefficients(t), scale)
        if on_step is not None:                              # the harness's per-step plots (generator/diffusion.py:648-674)
            on_step(i, x.reshape(B, L, 1))
    return x.reshape(B, L, 1)


def draw_ensemble_starts(guid: Guidance, n_groups: int, n_obj: int, n_steps: int, streams: Optional[Sequence[StartStream]] = None):
    """FPS starts of ``n_groups`` independent multi-object chains: every group has its own generator stream and consumes it as
    ``guided_sample_multi_object`` does - step after step, inside a step object after object (generator/diffusion.py:641-643).
    Returned as (n_steps, n_obj, n_groups, starts_per_call): the launch order of ``guided_multi_object_groups``."""
    streams = streams or [StartStream(guid.cfg.num_object_points, guid.cfg.sub_batch_size) for _ in range(n_groups)]
    out = np.zeros((n_steps, n_obj, n_groups, guid.starts_per_call), dtype=np.int64)
    for k in range(n_groups):
        for si in range(n_steps):
            for j in range(n_obj):
                out[si, j, k] = streams[k].call(guid.rows)
    return out


def guided_multi_object_groups(unet: Unet1d, guid: Guidance, sched: DDIMScheduler, mode: str, noise: torch.Tensor,
                               groups: Sequence[Sequence[int]], opt_objs: Sequence[str], streams: Optional[Sequence[StartStream]] = None,
                               predrawn: Optional[np.ndarray] = None, python_loop: bool = False) -> torch.Tensor:
    """Several independent ``guided_sample_multi_object`` chains (:637-647) in the same launches: chain k averages the guidance
    gradients of the objects ``groups[k]`` for objective ``opt_objs[k]`` (a guidance ensemble: n_obj dynamics-gradient
    evaluations per denoise step).  All groups have the same size.  Launch order of the gradient chains is object-major,
    (j, k) -> j * K + k, so the mean over j is one strided reduction for every group at once.  Returns (K, B, L, 1)."""
    K, n_obj, (B, L, _) = len(groups), len(groups[0]), noise.shape
    assert all(len(g) == n_obj for g in groups) and len(opt_objs) == K
    if any(o == 'convergence' for o in opt_objs):
        raise ValueError("the reference never runs the multi-object loop with 'convergence' (generator/diffusion.py:337)")
    dev = noise.device
    is3d = mode == 'point_3d'
    S = len(sched.timesteps)
    objectives = [make_objective(opt_objs[k], groups[k][j]) for j in range(n_obj) for k in range(K)]
    if is3d and predrawn is None:
        predrawn = draw_ensemble_starts(guid, K, n_obj, S, streams)
    scale = classifier_scale(mode, opt_objs[0], multi=True)
    if not python_loop:       # one library call: K chains x n_obj gradients each, gradient chains object-major
        out = engine.guided_chains_run(unet, guid, noise.reshape(B, L), K, n_obj, objectives, None,
                                       np.ascontiguousarray(predrawn) if is3d else None, [int(t) for t in sched.timesteps],
                                       [sched.coefficients(int(t)) for t in sched.timesteps], [scale] * K)
        return out.reshape(K, B, L, 1)
    x = noise.reshape(1, B, L).expand(K, -1, -1).contiguous().to(torch.float32)
    for si, t in enumerate(sched.timesteps):
        t = int(t)
        if si == 0 and K > 1:        # all K chains start from the same noise: one eps-net evaluation (see guided_chains)
            ts = torch.full((B,), t, dtype=torch.int32, device=dev)
            eps = unet.forward(x[0].reshape(B, L, 1), ts).reshape(1, B, L).expand(K, -1, -1).contiguous()
        else:
            ts = torch.full((K * B,), t, dtype=torch.int32, device=dev)
            eps = unet.forward(x.reshape(K * B, L, 1), ts).reshape(K, B, L)
        g = guid.grad(x.repeat(n_obj, 1, 1), t, objectives, None, predrawn[si].reshape(-1) if is3d else None)     # (n_obj*K, B, L)
        x = engine.ddim_guided_step(x, eps, g, n_obj, sched.coefficients(t), scale)
    return x.reshape(K, B, L, 1)
