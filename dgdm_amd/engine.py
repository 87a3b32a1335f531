"""Python handles over the C-ABI objects of libdgdm_hip.so.

PyTorch supplies device memory and the stream; every computation is a HIP kernel behind
``include/dgdm_hip.h``.  These classes are what the reference-shaped modules in
``dgdm_amd.generator`` / ``dgdm_amd.dynamics`` delegate to.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, dptr, lib, stream_ptr


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(dtype=torch.float32).contiguous()


class Unet1d:
    """``ConditionalUnet1D`` weights packed on the device (generator/diffusion_utils.py:123-285)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], down_dims: Sequence[int] = (128, 256), step_embed_dim: int = 32,
                 kernel_size: int = 5, n_groups: int = 8, contraction_dtype: str = "f32"):
        packed = _lib.PackedStateDict(state_dict)
        dd = (C.c_int32 * len(down_dims))(*down_dims)
        h = C.c_void_p()
        check(lib().dgdm_unet1d_create(C.byref(h), packed.array, packed.n, dd, len(down_dims), step_embed_dim, kernel_size, n_groups))
        self._h = h
        self.set_contraction_dtype(contraction_dtype)

    def set_contraction_dtype(self, dtype: str) -> None:
        """Arithmetic of the multi-channel convolutions: 'f32' (default, the parity path: float32-grade, every product as three f16 MFMA
        products on exactly scaled two-way split operands, csrc/unet.hip conv_mfma_f16x3; 'f32_f16x3' names the same form), 'f32_mfma' (the float32 MFMA chain of rounds 1-3) or 'bf16' (operands ROUNDED to bf16)."""
        codes = {"f32": 0, "bf16": 1, "f32_mfma": 2, "f32_f16x3": 3}
        if dtype not in codes:
            raise ValueError(f"contraction dtype {dtype!r} not supported")
        check(lib().dgdm_unet1d_set_contraction_dtype(self._h, codes[dtype]))
        self.contraction_dtype = dtype

    def effective_form(self, batch: int, num_points: int):
        """(arithmetic, batched) a forward of `batch` samples of `num_points` control points actually runs: arithmetic is 'f32_f16x3',
        'f32_mfma' (also what 'f32' falls back to where the split form's slabs do not fit the LDS: L = 44, 46) or 'bf16'; batched says
        whether the layer-by-layer form for large batches is used (same bits as the per-sample kernel)."""
        code = int(lib().dgdm_unet1d_effective_form(self._h, int(batch), int(num_points)))
        if code < 0:
            raise ValueError("bad batch / num_points")
        return {1: "bf16", 2: "f32_mfma", 3: "f32_f16x3"}[code & 15], bool(code & 16)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # module globals are None during interpreter shutdown
            lib().dgdm_unet1d_destroy(self._h)
            self._h = None

    def forward(self, sample: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
        """sample (B, L, 1) float32 cuda, timestep (B,) integer -> eps (B, L, 1)."""
        assert sample.dim() == 3 and sample.shape[-1] == 1, "input_dim must be 1 (generator/train.py:76)"
        x = _f32(sample)
        B, L, _ = x.shape
        t = timestep.to(device=x.device, dtype=torch.int32).expand(B).contiguous()
        out = torch.empty_like(x)
        check(lib().dgdm_unet1d_forward(self._h, dptr(x), dptr(t), dptr(out), B, L, stream_ptr()))
        return out


class UnetTrainer:
    """Training state of the eps-net on the device (csrc/unet_train.hip): parameters, gradients, Adam moments and the EMA copy, with
    one call per ``Diffusion.get_stats`` + backward + ``torch.optim.Adam`` step (generator/diffusion.py:126-177, 711-724)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], num_points: int, down_dims: Sequence[int] = (128, 256), step_embed_dim: int = 32,
                 kernel_size: int = 5, n_groups: int = 8, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self._keys = {k: tuple(v.shape) for k, v in state_dict.items()}
        packed = _lib.PackedStateDict(state_dict)
        dd = (C.c_int32 * len(down_dims))(*down_dims)
        h = C.c_void_p()
        check(lib().dgdm_unet_trainer_create(C.byref(h), packed.array, packed.n, num_points, dd, len(down_dims), step_embed_dim, kernel_size, n_groups,
                                             betas[0], betas[1], eps, weight_decay))
        self._h, self.num_points = h, num_points

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().dgdm_unet_trainer_destroy(self._h)
            self._h = None

    def _args(self, x0, noise, sqrt_abar, sqrt_1m_abar, timesteps):
        dev = torch.device("cuda", torch.cuda.current_device())
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()      # noqa: E731
        x0, noise = f(x0).reshape(x0.shape[0], -1), f(noise).reshape(x0.shape[0], -1)
        assert x0.shape[1] == self.num_points and noise.shape == x0.shape
        return x0, noise, f(sqrt_abar), f(sqrt_1m_abar), timesteps.detach().to(device=dev, dtype=torch.int64).contiguous()

    def step(self, x0, noise, sqrt_abar, sqrt_1m_abar, timesteps, lr: float, want_pred: bool = False, want_loss: bool = True):
        """One optimisation step; returns (loss or None, noise_pred (B, L, 1) or None)."""
        a = self._args(x0, noise, sqrt_abar, sqrt_1m_abar, timesteps)
        B = a[0].shape[0]
        pred = torch.empty_like(a[0]) if want_pred else None
        loss = C.c_float()
        check(lib().dgdm_unet_trainer_step(self._h, *[dptr(v) for v in a], B, float(lr), dptr(pred), C.byref(loss) if want_loss else None, stream_ptr()))
        return (float(loss.value) if want_loss else None), (pred.reshape(B, -1, 1) if want_pred else None)

    def forward_backward(self, x0, noise, sqrt_abar, sqrt_1m_abar, timesteps, total_samples: Optional[int] = None, backward: bool = True,
                         want_pred: bool = False):
        a = self._args(x0, noise, sqrt_abar, sqrt_1m_abar, timesteps)
        B = a[0].shape[0]
        pred = torch.empty_like(a[0]) if want_pred else None
        loss = C.c_float()
        check(lib().dgdm_unet_trainer_forward_backward(self._h, *[dptr(v) for v in a], B, int(total_samples or B), 1 if backward else 0, dptr(pred),
                                                       C.byref(loss), stream_ptr()))
        return float(loss.value), (pred.reshape(B, -1, 1) if want_pred else None)

    def gradient_count(self) -> int:
        return int(lib().dgdm_unet_trainer_gradient_count(self._h))

    def read_gradients(self) -> torch.Tensor:
        flat = torch.empty(self.gradient_count(), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
        check(lib().dgdm_unet_trainer_gradients(self._h, dptr(flat), flat.numel(), 0, 1.0, stream_ptr()))
        return flat

    def write_gradients(self, flat: torch.Tensor, scale: float = 1.0) -> None:
        check(lib().dgdm_unet_trainer_gradients(self._h, dptr(flat), flat.numel(), 1, float(scale), stream_ptr()))

    def apply(self, lr: float) -> None:
        check(lib().dgdm_unet_trainer_apply(self._h, float(lr), stream_ptr()))

    def ema_step(self, decay: float) -> None:
        check(lib().dgdm_unet_trainer_ema_step(self._h, float(decay), float(1 - decay), stream_ptr()))

    def steps(self) -> int:
        return int(lib().dgdm_unet_trainer_steps(self._h))

    def export(self, which: int = 0) -> Dict[str, torch.Tensor]:
        """which: 0 parameters, 1 gradients, 2 / 3 Adam's exp_avg / exp_avg_sq, 4 the EMA copy (host tensors, the U-Net's own keys)."""
        host = {k: torch.empty(shp, dtype=torch.float32) for k, shp in self._keys.items()}
        packed = _lib.PackedStateDict(host)
        check(lib().dgdm_unet_trainer_export(self._h, which, packed.array, packed.n))
        return {n.decode(): torch.from_numpy(a.copy()).reshape(self._keys[n.decode()]) for n, a in zip(packed.names, packed.keep)}

    def load(self, which: int, state_dict: Dict[str, torch.Tensor], adam_steps: int = -1) -> None:
        packed = _lib.PackedStateDict(state_dict)
        check(lib().dgdm_unet_trainer_import(self._h, which, packed.array, packed.n, adam_steps))


class Dynamics:
    """``ProfileForward2DModel`` (kind 2) / ``ProfileForward3DModel`` (kind 3) on the device."""

    def __init__(self, kind: int, state_dict: Dict[str, torch.Tensor], params_ch: int, object_ch: int = 0):
        packed = _lib.PackedStateDict(state_dict)
        h = C.c_void_p()
        check(lib().dgdm_dynamics_create(C.byref(h), kind, packed.array, packed.n, params_ch, object_ch))
        self._h, self.kind, self.params_ch, self.object_ch = h, kind, params_ch, object_ch

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # module globals are None during interpreter shutdown
            lib().dgdm_dynamics_destroy(self._h)
            self._h = None

    def forward2d(self, x_ctrl, x_ori, x_pos, timesteps, object_vertices) -> torch.Tensor:
        rows = x_ctrl.shape[0]
        a = [_f32(v) for v in (x_ctrl, x_ori, x_pos, timesteps, object_vertices)]
        out = torch.empty((rows, 3), dtype=torch.float32, device=a[0].device)
        check(lib().dgdm_dyn2d_forward(self._h, *[dptr(v) for v in a], dptr(out), rows, stream_ptr()))
        return out

    @staticmethod
    def _starts(s: torch.Tensor) -> np.ndarray:
        return np.ascontiguousarray(s.detach().cpu().numpy().astype(np.int64))

    def pointnet2(self, xyz: torch.Tensor, start_sa1: torch.Tensor, start_sa2: torch.Tensor) -> torch.Tensor:
        """xyz (rows, 3, N) -> (rows, 256)."""
        x = _f32(xyz)
        rows, _, N = x.shape
        s1, s2 = self._starts(start_sa1), self._starts(start_sa2)
        out = torch.empty((rows, 256), dtype=torch.float32, device=x.device)
        check(lib().dgdm_pointnet2_forward(self._h, dptr(x), s1.ctypes.data, s2.ctypes.data, dptr(out), rows, N, stream_ptr()))
        return out

    def forward3d(self, x_ctrl, x_ori, x_pos, timesteps, xyz, start_sa1, start_sa2) -> torch.Tensor:
        rows, _, N = xyz.shape
        a = [_f32(v) for v in (x_ctrl, x_ori, x_pos, timesteps, xyz)]
        s1, s2 = self._starts(start_sa1), self._starts(start_sa2)
        out = torch.empty((rows, 3), dtype=torch.float32, device=a[0].device)
        check(lib().dgdm_dyn3d_forward(self._h, *[dptr(v) for v in a], s1.ctypes.data, s2.ctypes.data, dptr(out), rows, N, stream_ptr()))
        return out


def debug_pointnet_indices(dyn: "Dynamics", cloud: torch.Tensor, perm: Optional[torch.Tensor] = None) -> Dict[str, np.ndarray]:
    """Test hook: the index decisions of the PointNet++ pipeline for one cloud (N, 3) - FPS sequences from every start index,
    sa1's 32-neighbour lists, sa2's first-64 lists for the candidate order `perm` (default: index order), crowded flags."""
    x = _f32(cloud)
    N = x.shape[0]
    pm = (torch.arange(N) if perm is None else perm).to(device=x.device, dtype=torch.int32).contiguous()
    mk = lambda *shape: torch.empty(shape, dtype=torch.int32, device=x.device)       # noqa: E731
    out = dict(fps512=mk(N, 512), fps128=mk(N, 128), fps128_flags=mk(N), ball1=mk(N, 32), ball2=mk(N, 64), ball2_count=mk(N), crowded=mk(N))
    check(lib().dgdm_debug_pointnet_indices(dyn._h, dptr(x), N, dptr(pm), pm.numel(), *[dptr(v) for v in out.values()], stream_ptr()))
    return {k: v.cpu().numpy() for k, v in out.items()}


def make_objective(name: str, object_index: int = 0) -> _lib.Objective:
    o = _lib.Objective()
    o.object = object_index
    check(lib().dgdm_objective_from_name(name.encode(), C.byref(o)))
    return o


class Guidance:
    """State of ``Diffusion.cond_fn`` for up to ``max_chains`` chains (generator/diffusion.py:473-504)."""

    def __init__(self, dyn: Dynamics, batch: int, grid_size: int, num_pos: int, ori_range: Sequence[float], max_chains: int,
                 num_train_timesteps: int, num_object_points: int, sub_batch_size: int = 0, max_objects: int = 8,
                 contraction_dtype: str = "f32"):
        cfg = _lib.GuidanceConfig(batch, grid_size, num_pos, float(ori_range[0]), float(ori_range[1]), max_chains,
                                  num_train_timesteps, sub_batch_size, num_object_points, max_objects)
        h = C.c_void_p()
        check(lib().dgdm_guidance_create(C.byref(h), dyn._h, C.byref(cfg)))
        self._h, self.dyn, self.cfg = h, dyn, cfg
        self.rows = int(lib().dgdm_guidance_rows(h))
        self.starts_per_call = int(lib().dgdm_guidance_starts_per_call(h))
        self.sweep_rows = batch * grid_size
        self.n_objects = 0
        self.set_contraction_dtype(contraction_dtype)

    def set_contraction_dtype(self, dtype: str) -> None:
        """Arithmetic of the trunk of cond_fn: 'f32' (default, = 'f32_f16x3': float32 operands as two exactly scaled f16 pieces, three f16
        MFMAs per product, float32 accumulation - float32-grade, csrc/trunk_f16l.hip), 'f32_mfma' (the k-ordered float32 MFMA chain,
        csrc/trunk.hip) or 'bf16' (operands ROUNDED to bf16, float32 accumulation)."""
        codes = {"f32": 0, "bf16": 1, "f32_mfma": 2, "f32_f16x3": 3}
        if dtype not in codes:
            raise ValueError(f"contraction dtype {dtype!r} not supported")
        check(lib().dgdm_guidance_set_contraction_dtype(self._h, codes[dtype]))
        self.contraction_dtype = dtype

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # module globals are None during interpreter shutdown
            lib().dgdm_guidance_destroy(self._h)
            self._h = None

    def set_objects(self, objects: torch.Tensor, wait: bool = False) -> None:
        """2-D: (n, V, 2); 3-D: (n, N, 3).  The table build (3-D) is enqueued on the current stream and the handle's build streams and
        the call returns: like every other call of the handle it is ordered by the stream - the coordinates are copied into the
        handle's pool by a device copy on the current stream before anything else reads them (so `objects` is only needed by that
        enqueued copy, the contract of any asynchronous call; torch's allocator keeps a freed block for work on the same stream), the
        build streams start behind everything already enqueued on the current stream (chains still reading the previous tables
        included), and whatever follows on the current stream waits for the build.  The host is then free to convert and sort the
        next chains' FPS draws while the tables are being built (`dgdm_guided_chains_run` waits for the build's read-back only when
        it needs it).  `wait=True` blocks until the tables exist."""
        o = _f32(objects)
        check(lib().dgdm_guidance_set_objects(self._h, dptr(o), o.shape[0], stream_ptr()))
        self._objects_ref = o                            # `o` may be a temporary: it is read by the copy still in flight
        if wait:
            torch.cuda.current_stream().synchronize()
        self.n_objects = o.shape[0]

    def debug_fps_path(self, mode):
        """Test hook: where a reference row's PointNet++ embedding comes from.  0 / False: default ((chain, s1)-group gather kernel
        until the objects have served more than 5 cond_fn calls, then the per-object embedding table and no gather at all); 5: the
        next set_objects builds the embedding tables right away; 3: always the group gather kernel; 2: per-row table kernel;
        1 / True: every row runs its own FPS(128); 4: like 0, and the next set_objects builds the crowded centres' features with
        global gathers.  All must agree bit for bit.  Returns which objects are admissible for the table path."""
        ok = (C.c_int32 * max(1, self.n_objects))()
        check(lib().dgdm_guidance_debug_fps_path(self._h, int(mode), ok))
        return [bool(v) for v in ok][:self.n_objects]

    def debug_partials(self, n_chains: int) -> torch.Tensor:
        """Test hook: per-tile partial sums of d objective / d z1 of the last grad() call -> (n_chains, B, tiles_per_finger, W1)."""
        tp, w = C.c_int32(), C.c_int32()
        check(lib().dgdm_guidance_debug_partials(self._h, n_chains, None, C.byref(tp), C.byref(w), stream_ptr()))
        out = torch.empty((n_chains, self.cfg.batch, tp.value, w.value), dtype=torch.float32, device="cuda")
        check(lib().dgdm_guidance_debug_partials(self._h, n_chains, dptr(out), None, None, stream_ptr()))
        return out

    def rowcoef(self, centers: torch.Tensor) -> np.ndarray:
        """'convergence' row coefficients of one chain (deltas_to_objective :445-452 applied per cond_fn call)."""
        c = np.ascontiguousarray(centers.detach().cpu().numpy().astype(np.int64))
        out = np.empty(self.rows, dtype=np.float32)
        sub = self.cfg.sub_batch_size if self.dyn.kind == 3 else 0
        check(lib().dgdm_convergence_rowcoef(c.ctypes.data, len(c), self.cfg.grid_size, self.cfg.num_pos, self.rows, sub, out.ctypes.data))
        return out

    def grad(self, x: torch.Tensor, timestep: int, objectives: Sequence[_lib.Objective], rowcoef: Optional[torch.Tensor] = None,
             starts: Optional[np.ndarray] = None) -> torch.Tensor:
        """x (n_chains, B, L) -> d sum(objective)/dx (n_chains, B, L)."""
        x = _f32(x)
        nc = x.shape[0]
        assert nc == len(objectives)
        arr = (_lib.Objective * nc)(*objectives)
        out = torch.empty_like(x)
        rc_ptr = dptr(rowcoef) if rowcoef is not None else None
        if self.dyn.kind == 2:
            check(lib().dgdm_dyn2d_guidance_grad(self._h, dptr(x), int(timestep), arr, rc_ptr, nc, dptr(out), stream_ptr()))
        else:
            assert starts is not None and starts.dtype == np.int64 and starts.size == nc * self.starts_per_call
            starts = np.ascontiguousarray(starts)
            check(lib().dgdm_dyn3d_guidance_grad(self._h, dptr(x), int(timestep), arr, rc_ptr, starts.ctypes.data, nc, dptr(out), stream_ptr()))
        return out

    def sweep(self, x: torch.Tensor, object_of_chain: Sequence[int], starts: Optional[np.ndarray] = None) -> torch.Tensor:
        """Orientation sweep of get_convergence_centers (:506-531): logits (n_chains, B*G, 3), row = g*B + b."""
        x = _f32(x)
        nc = x.shape[0]
        oc = (C.c_int32 * nc)(*object_of_chain)
        out = torch.empty((nc, self.sweep_rows, 3), dtype=torch.float32, device=x.device)
        sp = None
        if self.dyn.kind == 3:
            assert starts is not None and starts.dtype == np.int64 and starts.size == nc * 2 * self.sweep_rows
            starts = np.ascontiguousarray(starts)
            sp = starts.ctypes.data
        check(lib().dgdm_guidance_orientation_sweep(self._h, dptr(x), oc, sp, nc, dptr(out), stream_ptr()))
        return out


def ddim_guided_step(x: torch.Tensor, eps: torch.Tensor, grad: Optional[torch.Tensor], n_grad: int, coef: Tuple[float, float, float, float],
                     scale: float) -> torch.Tensor:
    """grad: None or (n_grad, *x.shape) stacked gradients whose mean guides the step."""
    x, eps = _f32(x), _f32(eps)
    out = torch.empty_like(x)
    g = _f32(grad) if grad is not None else None
    check(lib().dgdm_ddim_guided_step(dptr(x), dptr(eps), dptr(g), n_grad, dptr(out), x.numel(), *[float(c) for c in coef], float(scale), stream_ptr()))
    return out


def guided_chains_run(unet: "Unet1d", guid: "Guidance", noise: torch.Tensor, n_chains: int, n_grad: int, objectives: Sequence[_lib.Objective],
                      rowcoef: Optional[torch.Tensor], starts: Optional[np.ndarray], timesteps: Sequence[int],
                      coefs: Sequence[Tuple[float, float, float, float]], scales: Sequence[float]) -> torch.Tensor:
    """The whole guided denoise loop in one library call (dgdm_guided_chains_run): noise (B, L) -> (n_chains, B, L)."""
    x0 = _f32(noise).reshape(noise.shape[0], -1)
    B, L = x0.shape
    S = len(timesteps)
    assert len(objectives) == n_chains * n_grad and len(scales) == n_chains and len(coefs) == S
    arr = (_lib.Objective * len(objectives))(*objectives)
    ts = (C.c_int32 * S)(*[int(t) for t in timesteps])
    cf = (C.c_float * (4 * S))(*[float(v) for c in coefs for v in c])
    sc = (C.c_float * n_chains)(*[float(v) for v in scales])
    sp = None
    if guid.dyn.kind == 3:
        assert starts is not None and starts.dtype == np.int64 and starts.size == S * n_chains * n_grad * guid.starts_per_call
        starts = np.ascontiguousarray(starts)
        sp = starts.ctypes.data
    out = torch.empty((n_chains, B, L), dtype=torch.float32, device=x0.device)
    check(lib().dgdm_guided_chains_run(unet._h, guid._h, dptr(x0), n_chains, n_grad, arr, dptr(rowcoef) if rowcoef is not None else None, sp,
                                       ts, cf, sc, S, dptr(out), stream_ptr()))
    return out         # `starts` has been consumed: every step converts it into the handle's pinned staging buffer before it returns


def ddim_add_noise(x0: torch.Tensor, noise: torch.Tensor, sqrt_abar: float, sqrt_1m_abar: float) -> torch.Tensor:
    x0, noise = _f32(x0), _f32(noise)
    out = torch.empty_like(x0)
    check(lib().dgdm_ddim_add_noise(dptr(x0), dptr(noise), dptr(out), x0.numel(), float(sqrt_abar), float(sqrt_1m_abar), stream_ptr()))
    return out


def finger_decode_2d(samples: torch.Tensor, num_points: int = 200, scale: float = 0.03, offset: float = -0.015) -> torch.Tensor:
    """(B, L, 1) or (B, L) control values -> (B, 2 fingers, num_points, 2) spline points in metres; the defaults map sampler
    units [-1, 1] as dynamics/sim_test_mj.py:257-262 does before assets/finger_sampler.py:39-51."""
    s = _f32(samples).reshape(samples.shape[0], -1)
    out = torch.empty((s.shape[0], 2, num_points, 2), dtype=torch.float32, device=s.device)
    check(lib().dgdm_finger_decode_2d(dptr(s), s.shape[0], s.shape[1], int(num_points), float(scale), float(offset), dptr(out), stream_ptr()))
    return out


def finger_decode_3d(samples: torch.Tensor, sample_size: int = 25, scale: float = 0.05, offset: float = -0.05) -> torch.Tensor:
    """(B, 42, 1) or (B, 42) control values -> (B, 2 fingers, sample_size^2, 3) surface points in metres; the defaults map
    sampler units [-1, 1] as dynamics/sim_test_mj_3d.py:236-237 does before assets/finger_3d.py:60-81."""
    s = _f32(samples).reshape(samples.shape[0], -1)
    out = torch.empty((s.shape[0], 2, sample_size * sample_size, 3), dtype=torch.float32, device=s.device)
    check(lib().dgdm_finger_decode_3d(dptr(s), s.shape[0], s.shape[1], int(sample_size), float(scale), float(offset), dptr(out), stream_ptr()))
    return out


def prof_enable(on: bool) -> None:
    check(lib().dgdm_prof_enable(int(on)))


STAGES = ("trunk", "unet", "xobj", "tables", "guide_misc", "ddim")      # include/dgdm_hip.h DGDM_STAGE_*


def prof_read_stages() -> Dict[str, Tuple[int, float, float]]:
    """{stage: (bracketed regions, total ms, algorithmic work)} since prof_enable(True); clears the records."""
    out = {}
    for i, name in enumerate(STAGES):
        n, ms, wk = C.c_int64(), C.c_double(), C.c_double()
        check(lib().dgdm_prof_read_stage(i, C.byref(n), C.byref(ms), C.byref(wk)))
        out[name] = (n.value, ms.value, wk.value)
    return out


def prof_read() -> Tuple[int, float, float]:
    n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
    check(lib().dgdm_prof_read(C.byref(n), C.byref(ms), C.byref(fl)))
    return n.value, ms.value, fl.value
