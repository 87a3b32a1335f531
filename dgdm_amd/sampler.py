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

from . import _lib, engine
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


class TorchRng:
    """torch's CPU generator replayed by the library on the generator's own state blob (csrc/torch_rng.hip): the same numbers as
    ``torch.randint`` bit for bit, ~3x faster, callable from worker threads (ctypes releases the GIL) and able to SKIP draws.

    ``generator``: a ``torch.Generator``, or None for torch's global CPU generator - every operation then loads the generator's
    state, advances it and stores it back, so torch and this class can be used on the same generator alternately.
    ``seed``: a detached stream instead - the stream of ``torch.Generator().manual_seed(seed)`` without a torch object."""

    BYTES = 5056

    def __init__(self, generator: Optional[torch.Generator] = None, seed: Optional[int] = None, state: Optional[np.ndarray] = None):
        self._gen, self._detached = generator, seed is not None or state is not None
        self._blob = np.zeros(self.BYTES, dtype=np.uint8) if state is None else np.array(state, dtype=np.uint8, copy=True)
        if seed is not None:
            _lib.check(_lib.lib().dgdm_torch_rng_seed(self._blob.ctypes.data, self.BYTES, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def _load(self):
        if not self._detached:
            st = torch.get_rng_state() if self._gen is None else self._gen.get_state()
            self._blob = st.numpy().copy()

    def _store(self):
        if not self._detached:
            st = torch.from_numpy(self._blob.copy())
            torch.set_rng_state(st) if self._gen is None else self._gen.set_state(st)

    def detached_copy(self) -> "TorchRng":
        """A detached stream that starts where the generator stands now (for drawing ahead on a worker thread)."""
        self._load()
        return TorchRng(state=self._blob)

    def adopt(self, other: "TorchRng") -> None:
        """Puts the generator where the detached stream `other` stands."""
        self._blob = other._blob.copy()
        self._store()

    def randint(self, high: int, n: int, skip: bool = False) -> Optional[np.ndarray]:
        self._load()
        out = None if skip else np.empty(n, dtype=np.int64)
        _lib.check(_lib.lib().dgdm_torch_rng_randint(self._blob.ctypes.data, self.BYTES, int(high), int(n), None if skip else out.ctypes.data))
        self._store()
        return out

    def fps_starts(self, num_points: int, sub_batch_size: int, rows: int, n_calls: int = 1, skip: bool = False,
                   out: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """The draws of `n_calls` consecutive classifier calls over `rows` rows -> (n_calls, 2 * rows) int64 in the layout
        dgdm_dyn3d_guidance_grad takes, or None when skipped.  out: an int64 view of that shape to fill (rows contiguous; the
        calls may be strided - e.g. one chain's column of a [step][chain][2 * rows] array)."""
        self._load()
        stride = 0
        if skip:
            out = None
        elif out is None:
            out = np.empty((n_calls, 2 * rows), dtype=np.int64)
        else:
            assert out.dtype == np.int64 and out.shape == (n_calls, 2 * rows) and (n_calls == 0 or out.strides[1] == 8) and out.strides[0] % 8 == 0
            stride = out.strides[0] // 8 if n_calls > 1 else 0
        _lib.check(_lib.lib().dgdm_torch_rng_fps_starts(self._blob.ctypes.data, self.BYTES, int(num_points), max(1, int(sub_batch_size)), int(rows),
                                                       int(n_calls), None if out is None else out.ctypes.data, stride))
        self._store()
        return out


_ACTIVE_PLAN: Optional["StartPlan"] = None


class StartPlan:
    """Draws ahead.  The FPS start draws of a sweep do not depend on anything the GPU computes, only on the ORDER in which the
    reference's loops consume the generator; given that order as a list of jobs ``(rows, n_calls, keep)`` a worker thread makes the
    draws while the GPU runs the chains of earlier jobs.  Used as a context manager around code that draws from the GLOBAL generator
    through ``StartStream``: inside it those streams take their numbers from the plan (jobs must be asked for in plan order; a
    request that does not match cancels the plan and the stream continues synchronously from the right state).  On exit the global
    generator stands where the consumed jobs leave it - exactly as if every draw had been a ``torch.randint``."""

    def __init__(self, num_points: int, sub_batch_size: int, jobs: Sequence[Tuple[int, int, bool]], depth: int = 4):
        import queue
        import threading
        self.N, self.sub, self.jobs = int(num_points), int(sub_batch_size), [(int(r), int(n), bool(k)) for r, n, k in jobs]
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self._pos = 0
        self._rng_after: Optional[TorchRng] = None      # stream position after the last consumed job
        self.draw_seconds = 0.0

    def _put(self, item) -> None:
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.05)
                return
            except Exception:           # queue.Full
                continue

    def _work(self, rng: TorchRng):
        import time
        try:
            for rows, n, keep in self.jobs:
                if self._stop.is_set():
                    return
                t0 = time.perf_counter()
                arr = rng.fps_starts(self.N, self.sub, rows, n, skip=not keep)
                self.draw_seconds += time.perf_counter() - t0
                self._put((arr, TorchRng(state=rng._blob)))
        except BaseException as e:      # a bad blob / argument raises in the library: tell the consumer instead of leaving it waiting
            self._put(e)

    def __enter__(self):
        import threading
        global _ACTIVE_PLAN
        assert _ACTIVE_PLAN is None, "StartPlan contexts do not nest"
        self._base = TorchRng()                              # the global generator
        self._rng_after = self._base.detached_copy()
        self._thread = threading.Thread(target=self._work, args=(self._base.detached_copy(),), daemon=True)
        self._thread.start()
        _ACTIVE_PLAN = self
        return self

    def _close(self):
        global _ACTIVE_PLAN
        if _ACTIVE_PLAN is self:
            _ACTIVE_PLAN = None
            self._stop.set()
            while self._thread is not None and self._thread.is_alive():
                try:
                    self._q.get_nowait()
                except Exception:
                    pass
                self._thread.join(timeout=0.01)
            self._base.adopt(self._rng_after)

    def __exit__(self, *exc):
        self._close()
        return False

    def take(self, num_points: int, sub: int, rows: int, n_calls: int, keep: bool):
        """The next job's draws if it is the job asked for; otherwise cancels the plan (returns False)."""
        if self._pos < len(self.jobs) and (self.N, self.sub) == (int(num_points), int(sub)) and self.jobs[self._pos] == (int(rows), int(n_calls), bool(keep)):
            item = None
            while item is None:
                try:
                    item = self._q.get(timeout=0.5)
                except Exception:       # queue.Empty: still drawing - or the worker died without a word
                    if self._thread is None or not self._thread.is_alive():
                        try:
                            item = self._q.get_nowait()
                        except Exception:
                            item = RuntimeError("the draw-ahead worker ended without delivering this job")
            if isinstance(item, BaseException):      # fall back on the synchronous draw from the last consumed state
                self._close()
                return False, None
            arr, after = item
            self._pos += 1
            self._rng_after = after
            return True, arr
        self._close()
        return False, None


class StartStream:
    """FPS start indices in the reference's draw order.

    One classifier call on ``rows`` rows with sub-batch size ``sub`` draws, per sub-batch, ``torch.randint(0, N, (n,))``
    for sa1 and ``torch.randint(0, 512, (n,))`` for sa2 from the global CPU generator
    (pointnet2_utils.py:83 reached via generator/diffusion.py:495-498 and :524-526).  The numbers are made by the library's
    replay of that generator (``TorchRng``: identical to ``torch.randint``, tests/test_host_logic.py) on the generator's own state.

    ``generator``: a private ``torch.Generator`` instead of the global one; ``seed``: a detached stream (``pair_stream``);
    ``forced``: recorded draws (golden fixtures)."""

    def __init__(self, num_points: int, sub_batch_size: int, forced: Optional[Sequence[torch.Tensor]] = None,
                 generator: Optional[torch.Generator] = None, seed: Optional[int] = None):
        self.N, self.sub = int(num_points), int(sub_batch_size)
        self._forced = list(forced) if forced is not None else None
        self._rng = None if forced is not None else TorchRng(generator, seed)
        self._plannable = forced is None and generator is None and seed is None

    def _forced_call(self, rows: int) -> np.ndarray:
        out = np.empty(2 * rows, dtype=np.int64)
        for r0 in range(0, rows, self.sub):
            n = min(self.sub, rows - r0)
            for k, off in ((0, 2 * r0), (1, 2 * r0 + n)):
                s = self._forced.pop(0)
                assert s.shape == (n,), (s.shape, n)
                out[off:off + n] = s.numpy()
        return out

    def calls(self, rows: int, n_calls: int, keep: bool = True, out: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """The indices `n_calls` consecutive classifier calls over `rows` rows consume -> (n_calls, 2 * rows) int64, each row
        [sub-batch 0: sa1 x n0, sa2 x n0 | sub-batch 1: ...] - the layout dgdm_dyn3d_guidance_grad takes.  keep=False: the draws
        are consumed without being materialised (a rank replaying the stream past chains that belong to other ranks).
        out: where to put them (see TorchRng.fps_starts)."""
        arr = None
        if self._forced is not None:
            arr = np.stack([self._forced_call(rows) for _ in range(n_calls)]) if n_calls else np.empty((0, 2 * rows), np.int64)
            if not keep:
                return None
        elif self._plannable and _ACTIVE_PLAN is not None:
            ok, arr = _ACTIVE_PLAN.take(self.N, self.sub, rows, n_calls, keep)
            if ok and not keep:
                return None
            arr = arr if ok else None
        if arr is None:
            return self._rng.fps_starts(self.N, self.sub, rows, n_calls, skip=not keep, out=out)
        if out is not None:
            out[...] = arr
            return out
        return arr

    def call(self, rows: int) -> np.ndarray:
        """One classifier call: 2*rows indices in draw order."""
        return self.calls(rows, 1)[0]

    def skip(self, rows: int, n_calls: int = 1) -> None:
        self.calls(rows, n_calls, keep=False)


def pair_stream(num_points: int, sub_batch_size: int, seed: int, pair_index: int) -> StartStream:
    """A start stream of its own for pair `pair_index` (seed-derived), for workloads that have no reference draw order to keep
    (bench.py's synthetic pair batches): the draws of a pair depend on (seed, pair index) only, not on the rank that runs it or on
    how many ranks there are.  The stream of ``torch.Generator().manual_seed(...)`` with that seed."""
    return StartStream(num_points, sub_batch_size, seed=(int(seed) * 1_000_003 + int(pair_index)) & 0x7FFFFFFFFFFFFFFF)


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
                      keep: Optional[range] = None, streams: Optional[Sequence[StartStream]] = None, out: Optional[np.ndarray] = None,
                      pool=None):
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
    step = out if out is not None else np.empty((n_steps, len(keep), guid.starts_per_call), dtype=np.int64)
    assert step.shape == (n_steps, len(keep), guid.starts_per_call) and step.dtype == np.int64 and step.flags.c_contiguous
    def one(c, o, st, mine):
        k = c - keep.start
        if o == 'convergence':
            sw = st.calls(guid.sweep_rows, 1, keep=mine)
            if mine:
                sweep[k] = sw[0]
        st.calls(guid.rows, n_steps, keep=mine, out=step[:, k] if mine else None)

    if streams is not None:
        # independent streams: the chains' draws can be made side by side (`pool`: a concurrent.futures executor; the library call
        # releases the interpreter lock)
        jobs = [(c, o, streams[c], True) for c, (_, o) in enumerate(chains) if c in keep]
        if pool is not None and len(jobs) > 1:
            list(pool.map(lambda j: one(*j), jobs))
        else:
            for j in jobs:
                one(*j)
    else:
        for c, (_, o) in enumerate(chains):
            one(c, o, starts, c in keep)
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
            trace.append((eps.clone(), g.clone(), x.clone()))          # eps, gradient and the step's INPUT x
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
        st = starts.calls(guid.rows, S * n_obj).reshape(S, -1) if is3d else None      # per step object after object (:641-643): one run of draws
        out = engine.guided_chains_run(unet, guid, noise.reshape(B, L), 1, n_obj, objectives, None, st, [int(t) for t in sched.timesteps],
                                       [sched.coefficients(int(t)) for t in sched.timesteps], [scale])
        return out.reshape(B, L, 1)
    x = noise.reshape(B, L).contiguous().to(torch.float32)
    for i, t in enumerate(sched.timesteps):
        t = int(t)
        ts = torch.full((B,), t, dtype=torch.int32, device=dev)
        eps = unet.forward(x.reshape(B, L, 1), ts).reshape(B, L)
        st = starts.calls(guid.rows, n_obj).reshape(-1) if is3d else None      # object after object (:641-643)
        g = guid.grad(x.reshape(1, B, L).expand(n_obj, -1, -1).contiguous(), t, objectives, None, st)
        x = engine.ddim_guided_step(x, eps, g, n_obj, sched.coefficients(t), scale)
        if on_step is not None:                              # the harness's per-step plots (generator/diffusion.py:648-674)
            on_step(i, x.reshape(B, L, 1))
    return x.reshape(B, L, 1)


def draw_ensemble_starts(guid: Guidance, n_groups: int, n_obj: int, n_steps: int, streams: Optional[Sequence[StartStream]] = None,
                         out: Optional[np.ndarray] = None, pool=None):
    """FPS starts of ``n_groups`` independent multi-object chains: every group has its own generator stream and consumes it as
    ``guided_sample_multi_object`` does - step after step, inside a step object after object (generator/diffusion.py:641-643).
    Returned as (n_steps, n_obj, n_groups, starts_per_call): the launch order of ``guided_multi_object_groups``."""
    streams = streams or [StartStream(guid.cfg.num_object_points, guid.cfg.sub_batch_size) for _ in range(n_groups)]
    out = out if out is not None else np.empty((n_steps, n_obj, n_groups, guid.starts_per_call), dtype=np.int64)
    assert out.shape == (n_steps, n_obj, n_groups, guid.starts_per_call) and out.dtype == np.int64 and out.flags.c_contiguous

    def one(k):
        for si in range(n_steps):               # a group's stream is consumed step after step, object after object
            streams[k].calls(guid.rows, n_obj, out=out[si, :, k])

    if pool is not None and n_groups > 1:
        list(pool.map(one, range(n_groups)))
    else:
        for k in range(n_groups):
            one(k)
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
