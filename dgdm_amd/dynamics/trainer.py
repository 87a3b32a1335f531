"""``Trainer`` of the dynamics model with the reference's interface (dynamics/trainer.py:16-146) on the HIP path.

Parameters, gradients and the Adam state live in a ``DgdmTrainer2d`` / ``DgdmTrainer3d`` handle inside libdgdm_hip.so and one ``step``
is forward (BatchNorm in training mode) + MSE loss + backward + Adam on the GPU (csrc/train2d.hip; ``--fingers_3d``:
csrc/train3d.hip, PointNet++ in training mode as written).  The random draws are the reference's: ``torch.randn`` for the noise, then
``torch.randint`` for the timesteps, both from the CPU generator (trainer.py:68-74), and in 3-D the FPS start draws of every forward
(pointnet2_utils.py:83).  There is no CPU path.

The trained weights live in the library handle: ``Trainer.state_dict()`` / ``save_checkpoint()`` read them out; ``Trainer.model`` (the
``nn.Module`` the reference code pokes at for its key layout) keeps the INITIAL parameters unless ``sync_model()`` copies the trained
ones into it."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .. import dist as _dist
from .._lib import check, dptr, lib, stream_ptr
from ..scheduler import DDIMScheduler
from .profile_forward_2d import ProfileForward2DModel


class _Adam:
    """What dynamics/main.py reads of ``trainer.optimizer`` (:155): ``param_groups[0]['lr']``."""

    def __init__(self, lr: float, betas: Tuple[float, float], weight_decay: float):
        self.param_groups = [{"lr": lr, "initial_lr": lr, "betas": betas, "weight_decay": weight_decay, "eps": 1e-8}]


class _CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max, eta_min) in closed form (trainer.py:47)."""

    def __init__(self, optimizer: _Adam, T_max: int, eta_min: float):
        self.optimizer, self.T_max, self.eta_min, self.last_epoch = optimizer, T_max, eta_min, 0
        self.base_lr = optimizer.param_groups[0]["initial_lr"]

    def step(self) -> None:
        self.last_epoch += 1
        self.optimizer.param_groups[0]["lr"] = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]


class Trainer(object):
    def __init__(self, args):
        self.use_sub_batch = args.use_sub_batch
        self.sub_batch_size = args.sub_bs
        self.grid_size = args.grid_size
        self.learning_rate = args.learning_rate
        self.weight_decay = args.weight_decay
        self.num_epochs = args.num_epochs
        self.ckpt_path = args.checkpoint_path
        self.fingers_3d = args.fingers_3d
        self.gripperpts_dim = args.ctrlpts_dim
        self.object_vertices_dim = args.object_max_num_vertices if self.fingers_3d else 2 * args.object_max_num_vertices
        self.num_timesteps_per_batch = args.num_timesteps_per_batch
        if self.num_timesteps_per_batch != 1:
            # trainer.py:68-72 sizes the noise with num_timesteps_per_batch applied twice: only 1 is shape-consistent there
            raise NotImplementedError("num_timesteps_per_batch must be 1 (the value of dynamics/train_dynamics_2d.sh)")
        self.num_inference_steps = args.num_inference_steps
        self.noise_scheduler = DDIMScheduler(num_train_timesteps=args.num_train_timesteps, beta_schedule='squaredcos_cap_v2', clip_sample=True,
                                             prediction_type='epsilon')
        self.noise_scheduler.set_timesteps(self.num_inference_steps)
        self._h = None
        self.model: Optional[ProfileForward2DModel] = None
        # exact de-duplication of the time / object encoders over the rows (dgdm_trainer2d_set_groups); False = every row through both
        self.group_encoders = True
        # the next step's CPU-generator draws are made by a worker thread while the GPU runs this step (see _draw)
        self.draw_ahead = not self.fingers_3d      # 3-D: the FPS start draws of every forward follow on the same generator
        self._ahead = None

    # ------------------------------------------------------------------ model / optimizer (trainer.py:40-51)
    def create_model(self, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        if self.fingers_3d:
            from .profile_forward_3d import ProfileForward3DModel
            self.model = ProfileForward3DModel(output_ch=3, params_ch=self.gripperpts_dim)
        else:
            self.model = ProfileForward2DModel(output_ch=3, params_ch=self.gripperpts_dim, object_ch=self.object_vertices_dim)
        if state_dict is None and self.ckpt_path is not None:
            print('loading checkpoint from', self.ckpt_path)
            state_dict = torch.load(self.ckpt_path, map_location='cpu')
        if state_dict is not None:
            state_dict = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}
            self.model.load_state_dict(state_dict)
        self.optimizer = _Adam(self.learning_rate, (0.9, 0.95), self.weight_decay)
        self.lr_scheduler = _CosineAnnealingLR(self.optimizer, T_max=self.num_epochs, eta_min=1e-2 * self.learning_rate)
        if not torch.cuda.is_available():
            raise RuntimeError("dgdm_amd runs on an MI355X through libdgdm_hip.so; no GPU is visible and there is no CPU path")
        self._packed = _lib.PackedStateDict(self.model.plain_state_dict())
        h = C.c_void_p()
        if self.fingers_3d:
            check(lib().dgdm_trainer3d_create(C.byref(h), self._packed.array, self._packed.n, self.gripperpts_dim, self.object_vertices_dim,
                                              0.9, 0.95, 1e-8, float(self.weight_decay)))
        else:
            check(lib().dgdm_trainer2d_create(C.byref(h), self._packed.array, self._packed.n, self.gripperpts_dim, self.object_vertices_dim,
                                              0.9, 0.95, 1e-8, float(self.weight_decay)))
        self._h = h
        self._nbt0 = {k: int(v) for k, v in self.model.state_dict().items() if k.endswith('num_batches_tracked')}
        print('done')

    def _join_ahead(self):
        ahead, self._ahead = getattr(self, "_ahead", None), None
        if ahead is not None:
            ahead[0].join()              # a daemon thread running torch code must not outlive the interpreter's teardown

    def __del__(self):
        try:
            self._join_ahead()
            if getattr(self, "_h", None) and lib is not None:
                (lib().dgdm_trainer3d_destroy if self.fingers_3d else lib().dgdm_trainer2d_destroy)(self._h)
        except Exception:        # interpreter shutdown: module globals are already gone
            pass
        self._h = None

    # ------------------------------------------------------------------ one batch
    def _draw(self, rows: int, ahead: bool = True):
        """The reference's draws, in its order, from the CPU generator (trainer.py:68-74): ``torch.randn`` for the noise, then
        ``torch.randint`` for the timesteps.  16 M normals per 1.15 M-row step take as long on one host core as the GPU step itself,
        and they depend on nothing but the generator: after every step a worker thread draws the NEXT step's numbers (same row
        count) from a private generator that starts at the global generator's state; the next call adopts them - and moves the
        global generator to where those draws leave it - if the global state is still the one the worker started from (nobody else
        drew in between, e.g. a DataLoader reshuffle) and the row count matches; otherwise they are thrown away and drawn here."""
        pending, self._ahead = getattr(self, "_ahead", None), None
        out = None
        if pending is not None:
            th, box = pending
            th.join()
            if box.get("rows") == rows and "state_after" in box and torch.equal(torch.get_rng_state(), box["state_before"]):
                torch.set_rng_state(box["state_after"])
                out = (box["noise"], box["timesteps"])
        if out is None:
            noise = torch.randn((rows * self.num_timesteps_per_batch, self.gripperpts_dim))
            timesteps = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (rows,)).long()
            out = (noise, timesteps)
        if self.draw_ahead and ahead:
            self._start_draw_ahead(rows)
        return out

    def _start_draw_ahead(self, rows: int):
        import threading
        box = {"rows": rows, "state_before": torch.get_rng_state()}
        T, n, dim = self.noise_scheduler.config.num_train_timesteps, self.num_timesteps_per_batch, self.gripperpts_dim

        def work():
            g = torch.Generator()
            g.set_state(box["state_before"])
            # two pinned buffers, used alternately and kept while the shape stays the same: the previous draw may still be on its way to
            # the device (non-blocking copy) when this one is written
            pins = getattr(self, "_pins", None)
            if pins is None or pins[0].shape != (rows * n, dim):
                pins = self._pins = [torch.empty((rows * n, dim), pin_memory=torch.cuda.is_available()) for _ in range(2)]
                self._pin_turn = 0
            self._pin_turn ^= 1
            noise = pins[self._pin_turn]
            torch.randn((rows * n, dim), generator=g, out=noise)
            box["noise"] = noise
            box["timesteps"] = torch.randint(0, T, (rows,), generator=g).long()
            box["state_after"] = g.get_state()
        th = threading.Thread(target=work, daemon=True)
        th.start()
        self._ahead = (th, box)

    def _inputs(self, ctrl, score, input_ori, input_pos, object_vertices, drawn=None, ahead=True):
        dev = torch.device("cuda", torch.cuda.current_device())
        n = self.num_timesteps_per_batch
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32, non_blocking=t.is_pinned()).contiguous()       # noqa: E731
        ctrl_all, obj_all = f(ctrl.repeat(n, 1)), f(object_vertices.repeat(n, 1))
        ori_all, pos_all, score_all = f(input_ori.repeat(n, 1)), f(input_pos.repeat(n, 1)), f(score.repeat(n, 1))
        rows = ctrl_all.shape[0]
        noise, timesteps = drawn if drawn is not None else self._draw(rows, ahead)
        ac = self.noise_scheduler.alphas_cumprod[timesteps]
        sa, sb = f(ac ** 0.5), f((1 - ac) ** 0.5)                                       # DDIMScheduler.add_noise (diffusers 0.11.1)
        T = self.noise_scheduler.config.num_train_timesteps
        t = f(timesteps.float() / T)                                                       # rescale to [0,1] (:80)
        self._t_index = timesteps.to(device=dev, dtype=torch.int32).contiguous() if T <= 32 else None
        self._t_values = f(torch.arange(T).float() / T) if T <= 32 else None
        return ctrl_all, f(noise), sa, sb, t, ori_all, pos_all, obj_all, score_all, rows

    def _hint(self, lo: int, hi: int, rows_per_sample: Optional[int]):
        """Grouping hints for the rows [lo, hi) of the batch the next library call works on."""
        if not self.group_encoders:
            return
        g = _lib.TrainGroups()
        if self._t_index is not None:
            self._t_slice = self._t_index[lo:hi].contiguous()
            g.t_index_dev, g.t_values_dev, g.n_t = dptr(self._t_slice), dptr(self._t_values), int(self._t_values.numel())
        if rows_per_sample and rows_per_sample > 1 and lo % rows_per_sample == 0 and (hi - lo) % rows_per_sample == 0:
            g.rows_per_object = int(rows_per_sample)
        check(lib().dgdm_trainer2d_set_groups(self._h, C.byref(g)))

    def _run(self, ctrl, score, input_ori, input_pos, object_vertices, train: bool, rows_per_sample: Optional[int] = None, drawn=None):
        if self._h is None:
            raise RuntimeError("Trainer.create_model() has not been called")
        world, rank = _dist.world_rank()
        c, nz, sa, sb, t, o, p, ob, sc, rows = self._inputs(ctrl, score, input_ori, input_pos, object_vertices, drawn, ahead=train)
        lr = float(self.optimizer.param_groups[0]["lr"])
        loss = C.c_float()
        if world == 1:
            pred = torch.empty((rows, 3), dtype=torch.float32, device=c.device)
            self._hint(0, rows, rows_per_sample)
            check(lib().dgdm_trainer2d_step(self._h, dptr(c), dptr(nz), dptr(sa), dptr(sb), dptr(t), dptr(o), dptr(p), dptr(ob), dptr(sc), rows, lr,
                                            1 if train else 0, dptr(pred), C.byref(loss), stream_ptr()))
            return float(loss.value), pred
        # Data parallel, one process per GPU, with nn.DataParallel's semantics (trainer.py:41-43): the batch is cut into `world` chunks
        # like torch.chunk does for scatter, every replica normalises with ITS chunk's statistics, the loss is the mean over the whole
        # batch, the replicas' gradients add up (RCCL all-reduce) and every rank takes the same Adam step.  Every rank was handed the
        # whole batch and drew the whole batch's noise and timesteps from the synchronised CPU generator (dist.init_from_env).
        cs = -(-rows // world)
        lo, hi = min(rows, rank * cs), min(rows, (rank + 1) * cs)
        n = hi - lo
        # decided from (rows, world), which every rank knows - a rank raising alone would leave the others waiting in the all-reduce
        if train and any(min(rows, (r + 1) * cs) - min(rows, r * cs) == 1 for r in range(world)):
            raise ValueError("Expected more than 1 value per channel when training (a DataParallel chunk of one row)")
        cut = lambda v: v[lo:hi].contiguous()                                           # noqa: E731
        pred = torch.zeros((cs, 3), dtype=torch.float32, device=c.device)
        share = 0.0
        if train:
            flat = torch.zeros(int(lib().dgdm_trainer2d_gradient_count(self._h)), dtype=torch.float32, device=c.device)
            if n:
                self._hint(lo, hi, rows_per_sample)
                check(lib().dgdm_trainer2d_forward_backward(self._h, *[dptr(cut(v)) for v in (c, nz, sa, sb, t, o, p, ob, sc)], n, rows, dptr(pred),
                                                            C.byref(loss), stream_ptr()))
                check(lib().dgdm_trainer2d_gradients(self._h, dptr(flat), flat.numel(), 0, stream_ptr()))
                share = float(loss.value)
            flat = _dist.all_reduce_sum(flat)
            check(lib().dgdm_trainer2d_gradients(self._h, dptr(flat), flat.numel(), 1, stream_ptr()))
            check(lib().dgdm_trainer2d_apply(self._h, lr, stream_ptr()))
            # BatchNorm running statistics: every rank has just updated its own from ITS chunk; nn.DataParallel keeps replica 0's
            # (buffers of the other replicas are thrown away with them), and that is what rank 0 saves - so every rank evaluates with them
            run = torch.empty(8 * 512, dtype=torch.float32, device=c.device)
            check(lib().dgdm_trainer2d_running_stats(self._h, dptr(run), run.numel(), 0, stream_ptr()))
            run = _dist.broadcast_from_rank0(run)
            check(lib().dgdm_trainer2d_running_stats(self._h, dptr(run), run.numel(), 1, stream_ptr()))
        elif n:
            self._hint(lo, hi, rows_per_sample)
            check(lib().dgdm_trainer2d_step(self._h, *[dptr(cut(v)) for v in (c, nz, sa, sb, t, o, p, ob, sc)], n, lr, 0, dptr(pred), C.byref(loss),
                                            stream_ptr()))
            share = float(loss.value) * n / rows
        tail = torch.zeros((1, 3), dtype=torch.float32, device=c.device)
        tail[0, 0] = share
        got = _dist.all_gather_rows(torch.cat([pred, tail]))                            # [world, cs + 1, 3]
        full = torch.cat([got[r, :max(0, min(rows, (r + 1) * cs) - min(rows, r * cs))] for r in range(world)])
        return float(got[:, cs, 0].sum()), full

    # ------------------------------------------------------------------ 3-D (csrc/train3d.hip)
    def _run3d(self, ctrl, score, input_ori, input_pos, object_vertices, train: bool, drawn):
        """One forward (/ backward / optimizer step) on the rows handed in - a whole batch or one --use_sub_batch slice.  ctrl
        (rows, 3, L): only channel 1 is noised (trainer.py:68) and read (profile_forward_3d.py:77); object_vertices (rows, 3, N)."""
        if self._h is None:
            raise RuntimeError("Trainer.create_model() has not been called")
        world, rank = _dist.world_rank()
        dev = torch.device("cuda", torch.cuda.current_device())
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()       # noqa: E731
        noise, timesteps = drawn
        rows, N = ctrl.shape[0], object_vertices.shape[2]
        ac = self.noise_scheduler.alphas_cumprod[timesteps]
        T = self.noise_scheduler.config.num_train_timesteps
        # the FPS start draws of this forward: sa1 over the cloud's N points, then sa2 over sa1's 512 centres (pointnet2_utils.py:83)
        s1 = np.ascontiguousarray(torch.randint(0, N, (rows,), dtype=torch.long).numpy())
        s2 = np.ascontiguousarray(torch.randint(0, 512, (rows,), dtype=torch.long).numpy())
        lr = float(self.optimizer.param_groups[0]["lr"])
        loss = C.c_float()
        if world == 1:
            args = [f(ctrl[:, 1, :]), f(noise), f(ac ** 0.5), f((1 - ac) ** 0.5), f(timesteps.float() / T), f(input_ori), f(input_pos),
                    f(object_vertices.permute(0, 2, 1))]
            sc = f(score)
            pred = torch.empty((rows, 3), dtype=torch.float32, device=dev)
            check(lib().dgdm_trainer3d_step(self._h, *[dptr(v) for v in args], s1.ctypes.data, s2.ctypes.data, dptr(sc), rows, lr, 1 if train else 0, dptr(pred),
                                            C.byref(loss), stream_ptr()))
            return float(loss.value), pred
        # Data parallel, one process per GPU, nn.DataParallel's semantics (trainer.py:41-43 wraps the 3-D model too): the rows are cut into
        # `world` chunks as torch.chunk does for scatter, every replica's BatchNorm layers normalise with ITS chunk's statistics, the loss is
        # the mean over all rows, the replicas' gradients add up (RCCL all-reduce) and every rank takes the same Adam step; rank 0's
        # running statistics are the ones that survive (broadcast).  Every rank was handed all rows and made all draws - noise, timesteps
        # and the FPS starts - from the synchronised CPU generator (dist.init_from_env) and keeps its chunk's: the reference's replicas
        # draw their FPS starts from that one generator in thread order, which is not defined; chunk order is the single-process order.
        cs = -(-rows // world)
        lo, hi = min(rows, rank * cs), min(rows, (rank + 1) * cs)
        n = hi - lo
        if train and any(min(rows, (r + 1) * cs) - min(rows, r * cs) == 1 for r in range(world)):
            raise ValueError("Expected more than 1 value per channel when training (a DataParallel chunk of one row)")
        args = [f(v[lo:hi]) for v in (ctrl[:, 1, :], noise, ac ** 0.5, (1 - ac) ** 0.5, timesteps.float() / T, input_ori, input_pos)]
        args.append(f(object_vertices[lo:hi].permute(0, 2, 1)))
        sc = f(score[lo:hi])
        a1, a2 = np.ascontiguousarray(s1[lo:hi]), np.ascontiguousarray(s2[lo:hi])
        pred = torch.zeros((cs, 3), dtype=torch.float32, device=dev)
        share = 0.0
        if train:
            flat = torch.zeros(int(lib().dgdm_trainer3d_gradient_count(self._h)), dtype=torch.float32, device=dev)
            if n:
                check(lib().dgdm_trainer3d_forward_backward(self._h, *[dptr(v) for v in args], a1.ctypes.data, a2.ctypes.data, dptr(sc), n, rows, dptr(pred),
                                                            C.byref(loss), stream_ptr()))
                check(lib().dgdm_trainer3d_gradients(self._h, dptr(flat), flat.numel(), 0, stream_ptr()))
                share = float(loss.value)
            flat = _dist.all_reduce_sum(flat)
            check(lib().dgdm_trainer3d_gradients(self._h, dptr(flat), flat.numel(), 1, stream_ptr()))
            check(lib().dgdm_trainer3d_apply(self._h, lr, stream_ptr()))
            run = torch.empty(int(lib().dgdm_trainer3d_running_stats_count(self._h)), dtype=torch.float32, device=dev)
            check(lib().dgdm_trainer3d_running_stats(self._h, dptr(run), run.numel(), 0, stream_ptr()))
            run = _dist.broadcast_from_rank0(run)
            check(lib().dgdm_trainer3d_running_stats(self._h, dptr(run), run.numel(), 1, stream_ptr()))
        elif n:
            check(lib().dgdm_trainer3d_step(self._h, *[dptr(v) for v in args], a1.ctypes.data, a2.ctypes.data, dptr(sc), n, lr, 0, dptr(pred), C.byref(loss),
                                            stream_ptr()))
            share = float(loss.value) * n / rows
        tail = torch.zeros((1, 3), dtype=torch.float32, device=dev)
        tail[0, 0] = share
        got = _dist.all_gather_rows(torch.cat([pred, tail]))                            # [world, cs + 1, 3]
        full = torch.cat([got[r, :max(0, min(rows, (r + 1) * cs) - min(rows, r * cs))] for r in range(world)])
        return float(got[:, cs, 0].sum()), full

    def _draw3d(self, rows: int):
        """trainer.py:68-73 for --fingers_3d: randn for channel 1's noise (the zeros around it draw nothing), then the timesteps."""
        noise = torch.randn((rows * self.num_timesteps_per_batch, 1, self.gripperpts_dim))[:, 0, :]
        return noise, torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (rows,)).long()

    def _batch3d(self, ctrl, score, input_ori, input_pos, object_vertices, train: bool):
        n = ctrl.shape[0]
        noise, timesteps = self._draw3d(n)
        if not self.use_sub_batch:
            return self._run3d(ctrl, score, input_ori, input_pos, object_vertices, train, (noise, timesteps))
        losses, preds = [], []
        for i in range(0, n, self.sub_batch_size):
            sl = slice(i, i + self.sub_batch_size)
            loss, pred = self._run3d(ctrl[sl], score[sl], input_ori[sl], input_pos[sl], object_vertices[sl], train, (noise[sl], timesteps[sl]))
            losses.append(loss)
            preds.append(pred)
        return sum(losses) / (n / self.sub_batch_size), torch.cat(preds, dim=0)

    def step(self, ctrl, score, input_ori=None, input_pos=None, object_vertices=None, rows_per_sample: Optional[int] = None):
        """trainer.py:53-103: returns (loss.item(), pred.detach()).  rows_per_sample (not in the reference): the caller's promise that
        `object_vertices` holds runs of that many identical rows - dynamics/main.py builds its batches so - which lets the object
        encoder run once per sample."""
        if self.fingers_3d:
            return self._batch3d(ctrl, score, input_ori, input_pos, object_vertices, True)
        if not self.use_sub_batch:
            return self._run(ctrl, score, input_ori, input_pos, object_vertices, True, rows_per_sample)
        # --use_sub_batch (trainer.py:81-94): the draws once for the whole batch, then one optimizer step per slice of sub_bs rows;
        # returns the mean of the slices' losses (as the reference weighs them) and all predictions
        losses, preds, n = [], [], ctrl.shape[0]
        noise, timesteps = self._draw(n)
        for i in range(0, n, self.sub_batch_size):
            sl = slice(i, i + self.sub_batch_size)
            loss, pred = self._run(ctrl[sl], score[sl], input_ori[sl], input_pos[sl], object_vertices[sl], True, None, (noise[sl], timesteps[sl]))
            losses.append(loss)
            preds.append(pred)
        return sum(losses) / (n / self.sub_batch_size), torch.cat(preds, dim=0)

    def inference(self, ctrl, score, input_ori=None, input_pos=None, object_vertices=None, rows_per_sample: Optional[int] = None):
        """trainer.py:108-146 (eval mode, no update): returns (pred, loss)."""
        if self.fingers_3d:
            loss, pred = self._batch3d(ctrl, score, input_ori, input_pos, object_vertices, False)
            return pred, loss
        if not self.use_sub_batch:
            loss, pred = self._run(ctrl, score, input_ori, input_pos, object_vertices, False, rows_per_sample)
            return pred, loss
        # --use_sub_batch (trainer.py:132-141): the draws once for the whole batch, one forward per slice of sub_bs rows, the slices'
        # losses summed and divided by rows / sub_bs (so a ragged last slice weighs like a full one, as in the reference)
        losses, preds, n = [], [], ctrl.shape[0]
        noise, timesteps = self._draw(n, ahead=False)
        for i in range(0, n, self.sub_batch_size):
            sl = slice(i, i + self.sub_batch_size)
            loss, pred = self._run(ctrl[sl], score[sl], input_ori[sl], input_pos[sl], object_vertices[sl], False, None, (noise[sl], timesteps[sl]))
            losses.append(loss)
            preds.append(pred)
        return torch.cat(preds, dim=0), sum(losses) / (n / self.sub_batch_size)

    # ------------------------------------------------------------------ state
    def _export(self, which: int) -> Dict[str, torch.Tensor]:
        sd = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        packed = _lib.PackedStateDict({k: v for k, v in sd.items() if v.dtype == torch.float32})
        check((lib().dgdm_trainer3d_export if self.fingers_3d else lib().dgdm_trainer2d_export)(self._h, which, packed.array, packed.n))
        out = {}
        for name, arr in zip(packed.names, packed.keep):
            out[name.decode()] = torch.from_numpy(arr.copy()).reshape(sd[name.decode()].shape)
        if which == 0:
            steps = int((lib().dgdm_trainer3d_steps if self.fingers_3d else lib().dgdm_trainer2d_steps)(self._h))
            for k, v0 in self._nbt0.items():
                out[k] = torch.tensor(v0 + steps, dtype=torch.long)
        else:
            out = {k: v for k, v in out.items() if 'running_' not in k}
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The trained model's state_dict (host tensors), keys as the reference's bare module gives them."""
        return self._export(0)

    def sync_model(self):
        """Copies the trained parameters and running statistics from the library handle into ``self.model`` and returns it."""
        self.model.load_state_dict(self.state_dict())
        return self.model

    def gradients(self) -> Dict[str, torch.Tensor]:
        return self._export(1)

    def save_checkpoint(self, checkpoint_path):
        """trainer.py:105-106: the DataParallel-wrapped model's state_dict ('module.' prefix)."""
        torch.save({'module.' + k: v for k, v in self.state_dict().items()}, checkpoint_path)
