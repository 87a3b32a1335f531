"""``Diffusion`` with the reference's constructor and sampling methods (generator/diffusion.py:35-709), running on
libdgdm_hip.so.

What is kept: constructor keywords, ``noise_pred_net`` / ``ema_nets`` naming (so Lightning checkpoints load),
``cond_fn``, ``deltas_to_objective``, ``get_convergence_centers``, ``guided_sample``, ``guided_sample_multi_object``,
``validation_step``, the ``SCALE_*`` constants and the exceptions.  What is different by design:

* the objects of ``guided_sample`` are advanced as one batch of chains (dgdm_amd/sampler.py) instead of one after
  the other; the FPS start draws are replayed in the reference's order so the numbers are the same;
* the MuJoCo / Ray simulation that follows each loop in the reference (:577-580, :678-683) is not part of this package
  (SURVEY.md §2 #10): the final samples are returned and, if ``save_dir`` is given, written as ``.npy`` where the reference
  starts its simulation.  The rest of the harness is here (``artefacts.py``): the per-step PNG dumps (``val_vis/``,
  ``val_vis_noise/``, ``vis_guided/<tag>/allobj_*``) and - when a simulator callable with the reference's ``sim_test_batch``
  signature is given as ``Diffusion(simulator=...)`` - the unguided / guided / multi-object tables, written as JSON under
  ``tables/`` where the reference calls ``logger.log_table`` (a wandb logger passed as ``table_logger`` receives them too);
* Lightning is not required: the class is a plain ``nn.Module`` with the few hooks ``generator/train.py`` uses.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import dist as ddist
from .. import engine, sampler
from . import artefacts
from ..dynamics._backed import HipBacked
from ..dynamics import metrics
from ..dynamics.metrics import metric2objective, objective_directions     # noqa: F401  (metric2objective: reference import)
from ..sampler import SCALE_2D, SCALE_2D_CONV, SCALE_3D, SCALE_3D_CONV, StartStream     # noqa: F401  (reference exports)

OBJECTIVE_SWEEP = ['convergence', 'shift_up', 'shift_down', 'shift_left', 'shift_right', 'rotate_clockwise',
                   'rotate_counterclockwise', 'rotate', 'clockwise_up', 'clockwise_left', 'counterclockwise_up',
                   'counterclockwise_left']     # generator/diffusion.py:307


class _Adam:
    """What callers read of ``torch.optim.Adam(self.ema_nets.parameters(), lr)`` (generator/diffusion.py:712): the param group."""

    def __init__(self, lr: float):
        self.param_groups = [{"lr": lr, "initial_lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False}]


class _CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR in closed form (generator/diffusion.py:713; Lightning steps it once per epoch)."""

    def __init__(self, optimizer: _Adam, T_max: int, eta_min: float):
        self.optimizer, self.T_max, self.eta_min, self.last_epoch = optimizer, T_max, eta_min, 0
        self.base_lr = optimizer.param_groups[0]["initial_lr"]

    def step(self) -> None:
        import math
        self.last_epoch += 1
        self.optimizer.param_groups[0]["lr"] = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]


class EMAModel:
    """The decay schedule of ``diffusers.training_utils.EMAModel`` (0.11.1; generator/diffusion.py:83-87, 716-720) around the EMA copy
    the library handle keeps: ``decay = 0`` while ``step = max(0, optimization_step - update_after_step - 1) <= 0``, else
    ``1 - (1 + step / inv_gamma) ** -power`` clamped to [min_value, max_value]; ``step()`` = ema * decay + (1 - decay) * param on the
    device.  ``averaged_model`` is what the reference saves as ``state_dict['ema_model']``."""

    def __init__(self, trainer: "engine.UnetTrainer", update_after_step: int = 0, inv_gamma: float = 1.0, power: float = 2 / 3,
                 min_value: float = 0.0, max_value: float = 0.9999):
        self._trainer = trainer
        self.update_after_step, self.inv_gamma, self.power, self.min_value, self.max_value = update_after_step, inv_gamma, power, min_value, max_value
        self.decay, self.optimization_step = 0.0, 0

    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.update_after_step - 1)
        value = 1 - (1 + step / self.inv_gamma) ** -self.power
        if step <= 0:
            return 0.0
        return max(self.min_value, min(value, self.max_value))

    def step(self, new_model=None) -> None:
        self.decay = self.get_decay(self.optimization_step)
        self._trainer.ema_step(self.decay)
        self.optimization_step += 1

    @property
    def averaged_model(self) -> Dict[str, torch.Tensor]:
        return self._trainer.export(4)


class Diffusion(nn.Module):
    def __init__(self, noise_pred_net, noise_scheduler, num_inference_steps: int, num_epochs: int = 10000, mode: str = "point",
                 input_dim: int = 1, num_points: int = 10, H: int = 32, W: int = 32, learning_rate: float = 1e-4,
                 lr_warmup_steps: int = 0, ema_power: float = 0.75, ema_update_after_step: int = 0, num_timesteps_per_batch: int = 1,
                 action_groups: Optional[Dict[str, slice]] = None, float32_matmul_precision: str = "high", class_cond: bool = False,
                 classifier_model=None, grid_size: int = 360, num_pos: int = 5, object_vertices: Optional[torch.Tensor] = None,
                 object_ids: Optional[List[int]] = None, num_cpus: int = 32, sub_batch_size: int = 1024, pts_x_dim: int = 7,
                 pts_z_dim: int = 3, render_video: bool = False, seed: int = 0, contraction_dtype: str = "f32", simulator=None,
                 render_plots: bool = True, table_logger=None):
        super().__init__()
        if contraction_dtype not in ("f32", "bf16", "f32_mfma", "f32_f16x3"):
            raise ValueError(f"contraction dtype {contraction_dtype!r} not supported")
        # not a reference argument: 'bf16' runs the trunk / eps-net / sa3 contractions with bf16 operands (DESIGN_HISTORY.md 4.6)
        self.contraction_dtype = contraction_dtype
        # not reference arguments either: the simulator callable (sim_test_batch / sim_test_batch_3d signature) that feeds the
        # harness tables, whether the per-step PNGs are drawn (they cost a device->host copy per step, as in the reference), and
        # an optional wandb-style logger that receives the tables besides the JSON files
        self.simulator, self.render_plots, self.table_logger = simulator, render_plots, table_logger
        self.current_epoch = 0
        if mode not in ("point", "point_3d"):
            raise ValueError('model type not supported')
        self.ema_nets = nn.ModuleDict({"noise_pred_net": noise_pred_net})
        self.mode, self.input_dim, self.num_points = mode, input_dim, num_points
        self.pts_x_dim, self.pts_z_dim, self.H, self.W = pts_x_dim, pts_z_dim, H, W
        self.learning_rate, self.lr_warmup_steps, self.num_epochs = learning_rate, lr_warmup_steps, num_epochs
        self.ema_power, self.ema_update_after_step = ema_power, ema_update_after_step
        self.noise_scheduler, self.num_inference_steps = noise_scheduler, num_inference_steps
        self.num_timesteps_per_batch = num_timesteps_per_batch
        self.action_groups = action_groups if action_groups is not None else {}
        self.noise_scheduler.set_timesteps(num_inference_steps)
        self.class_cond, self.seed = class_cond, seed
        self.save_dir: Optional[str] = None
        self.last_samples: Dict[str, np.ndarray] = {}
        self.last_geometry: Dict[str, Any] = {}            # decoded finger curves / surfaces of the same batches (engine.finger_decode_*)
        if class_cond:
            self.classifier_model = classifier_model
            self.grid_size, self.num_pos = grid_size, num_pos
            self.object_vertices, self.object_ids = object_vertices, object_ids
            self.num_cpus, self.sub_batch_size, self.render_video = num_cpus, sub_batch_size, render_video
            self.use_sub_batch = mode == 'point_3d'
            thr, std = ([0.02, 0.001, 0.001], [0.0312, 0.0016, 0.0026]) if mode == 'point_3d' else \
                       ([0.03, 0.002, 0.003], [0.0565, 0.0026, 0.0047])
            self.threshold, self.std = torch.tensor(thr), torch.tensor(std)
            self.threshold_std = self.threshold / self.std
        self._guidance: Dict[Any, engine.Guidance] = {}

    # ------------------------------------------------------------------ plumbing
    @property
    def noise_pred_net(self):
        return self.ema_nets["noise_pred_net"]

    @property
    def device(self) -> torch.device:
        return next(self.noise_pred_net.parameters()).device

    def _net(self):
        h = self.noise_pred_net.handle()
        if getattr(h, "contraction_dtype", "f32") != self.contraction_dtype:
            h.set_contraction_dtype(self.contraction_dtype)
        return h

    def _dyn(self):
        m = self.classifier_model
        return m.module if hasattr(m, "module") else m            # nn.DataParallel wrapper of generator/train.py:86,88

    def clean_grad(self):
        for p in self.classifier_model.parameters():
            if p.grad is not None:
                p.grad.zero_()

    def load_state_dict(self, state_dict: Mapping[str, Any], strict: bool = True):
        """Accepts a Lightning checkpoint ``state_dict`` (keys ``ema_nets.noise_pred_net.*``, optional nested
        ``ema_model`` and ``_orig_mod.`` prefixes from torch.compile; reference :730-748)."""
        flat = {k.replace("_orig_mod.", ""): v for k, v in state_dict.items() if k != "ema_model"}
        out = super().load_state_dict(flat, strict=False)
        # nn.Module.load_state_dict copies into the children's parameters in place (it never calls the children's own
        # load_state_dict), so their packed device copies and every guidance handle bound to them are stale now
        for m in self.modules():
            if isinstance(m, HipBacked):
                m.invalidate()
        self._guidance.clear()
        self._unet_trainer = None          # a training handle built on the previous weights is stale
        return out

    def _guidance_for(self, batch: int, ori_range: Sequence[float], objects: torch.Tensor, max_chains: int) -> engine.Guidance:
        """Guidance handle for (B, ori_range) with `objects` as its bank."""
        dyn = self._dyn().handle()
        npts = objects.shape[1]
        key = (batch, float(ori_range[0]), float(ori_range[1]), npts)
        g = self._guidance.get(key)
        if g is None or g.dyn is not dyn or g.cfg.max_chains < max_chains or g.cfg.max_objects < objects.shape[0]:
            g = engine.Guidance(dyn, batch, self.grid_size, self.num_pos, ori_range, max(max_chains, 1),
                                self.noise_scheduler.config.num_train_timesteps, npts,
                                self.sub_batch_size if self.mode == 'point_3d' else 0, max_objects=max(objects.shape[0], 1),
                                contraction_dtype=self.contraction_dtype)
            self._guidance[key] = g
            g._bank = None
        if g._bank is None or g._bank.shape != objects.shape or not torch.equal(g._bank, objects.detach().cpu()):
            g.set_objects(objects.to(self.device))
            g._bank = objects.detach().cpu().clone()
        return g

    # ------------------------------------------------------------------ (f) rank 4: training the eps-net (generator/diffusion.py:126-177, 711-724)
    def configure_optimizers(self):
        """torch.optim.Adam(lr) + CosineAnnealingLR(T_max=num_epochs, eta_min=0) (:711-714); the optimiser state itself (both moments,
        the step count) lives in the library handle, these objects carry what callers read: ``param_groups[0]['lr']``, ``get_last_lr()``."""
        self.optimizer = _Adam(self.learning_rate)
        self.lr_scheduler = _CosineAnnealingLR(self.optimizer, T_max=self.num_epochs, eta_min=0.0)
        return [self.optimizer], [self.lr_scheduler]

    def _trainer(self) -> engine.UnetTrainer:
        if getattr(self, "_unet_trainer", None) is None:
            if not torch.cuda.is_available():
                raise RuntimeError("dgdm_amd runs on an MI355X through libdgdm_hip.so; no GPU is visible and there is no CPU path")
            net = self.noise_pred_net
            self._unet_trainer = engine.UnetTrainer(net.plain_state_dict(), self.num_points, net.down_dims, net.dsed, net.kernel_size, net.n_groups)
            self.ema = EMAModel(self._unet_trainer, power=self.ema_power, update_after_step=self.ema_update_after_step)
            self._trained = False
        return self._unet_trainer

    def _training_draws(self, tensor_data):
        """The draws of get_stats in its order and from its generator (:134-142: torch.randn then torch.randint on self.device - the
        device generator of torch, which PyTorch-ROCm provides as it is), and DDIMScheduler.add_noise's two coefficients per sample."""
        dev = self.device
        data = tensor_data.to(device=dev, non_blocking=True)
        n = data.shape[0] * self.num_timesteps_per_batch
        x0 = data.repeat(self.num_timesteps_per_batch, 1, 1)
        noise = torch.randn((n, self.num_points, self.input_dim), device=dev)
        timesteps = torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (n,), device=dev).long()
        ac = self.noise_scheduler.alphas_cumprod.to(dev)[timesteps]
        return x0, noise, ac ** 0.5, (1 - ac) ** 0.5, timesteps

    def get_stats(self, tensor_data) -> Dict[str, Any]:
        """:126-166 - the loss of one batch (forward only: nothing is updated; ``training_step`` is the call that trains)."""
        if not hasattr(self, "lr_scheduler"):
            self.configure_optimizers()
        loss, _ = self._trainer().forward_backward(*self._training_draws(tensor_data), backward=False)
        return {"loss": torch.tensor(loss), "lr": self.lr_scheduler.get_last_lr()[0]}

    def training_step(self, tensor_data, batch_idx):
        """:168-177 plus what Lightning's automatic optimisation does with the returned loss: backward and one optimizer.step() - one
        library call (csrc/unet_train.hip).  Under a process group (torchrun): Lightning's DDP semantics - every rank is handed ITS
        batch, the ranks' gradients are averaged (RCCL all-reduce), every rank takes the same Adam step."""
        if not hasattr(self, "lr_scheduler"):
            self.configure_optimizers()
        tr = self._trainer()
        lr = float(self.optimizer.param_groups[0]["lr"])
        world, _ = ddist.world_rank()
        args = self._training_draws(tensor_data)
        if world == 1:
            loss, _ = tr.step(*args, lr)
        else:
            loss, _ = tr.forward_backward(*args)
            tr.write_gradients(ddist.all_reduce_sum(tr.read_gradients()), 1.0 / world)
            tr.apply(lr)
        self._trained = True
        self.last_stats = {"train/loss": loss, "train/lr": self.lr_scheduler.get_last_lr()[0]}
        return torch.tensor(loss)

    def on_train_batch_end(self, outputs=None, batch=None, batch_idx=None) -> None:
        """:716-724 - EMAModel.step on the just-updated parameters."""
        self._trainer()
        self.ema.step()
        self.last_stats = dict(getattr(self, "last_stats", {}), **{"train/ema_decay": self.ema.decay})

    def sync_model(self) -> None:
        """Copies the trained parameters from the library handle into ``noise_pred_net`` (the module the sampling loops read, :120-124)."""
        if getattr(self, "_unet_trainer", None) is not None and getattr(self, "_trained", False):
            self.noise_pred_net.load_state_dict(self._unet_trainer.export(0))
            self._trained = False

    def checkpoint(self, epoch: int = 0, global_step: int = 0) -> Dict[str, Any]:
        """A Lightning-shaped checkpoint: ``state_dict`` (keys ``ema_nets.noise_pred_net.*``) with the nested ``ema_model`` the reference
        adds in on_save_checkpoint (:750-753, keys ``noise_pred_net.*``), Adam's state in torch.optim's own layout, the scheduler."""
        self.sync_model()
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items()}
        ck: Dict[str, Any] = {"epoch": epoch, "global_step": global_step, "state_dict": sd}
        tr = getattr(self, "_unet_trainer", None)
        if tr is not None:
            sd["ema_model"] = {"noise_pred_net." + k: v for k, v in tr.export(4).items()}
            names = [k for k, _ in self.noise_pred_net.named_parameters()]
            m, v = tr.export(2), tr.export(3)
            st = {i: {"step": torch.tensor(float(tr.steps())), "exp_avg": m[k], "exp_avg_sq": v[k]} for i, k in enumerate(names)}
            ck["optimizer_states"] = [{"state": st, "param_groups": [dict(self.optimizer.param_groups[0], params=list(range(len(names))))]}]
            ck["lr_schedulers"] = [{"T_max": self.lr_scheduler.T_max, "eta_min": self.lr_scheduler.eta_min, "last_epoch": self.lr_scheduler.last_epoch,
                                    "base_lrs": [self.lr_scheduler.base_lr], "_last_lr": self.lr_scheduler.get_last_lr()}]
            ck["ema"] = {"optimization_step": self.ema.optimization_step, "decay": self.ema.decay}
        return ck

    def load_checkpoint(self, ck: Mapping[str, Any]) -> None:
        """Resume: parameters (and, when present, the EMA copy, Adam's moments / step count and the schedule position)."""
        self.load_state_dict(ck["state_dict"] if "state_dict" in ck else ck)
        self._unet_trainer = None
        self.global_step = int(ck.get("global_step", 0))
        if "optimizer_states" not in ck:
            return
        if not hasattr(self, "lr_scheduler"):
            self.configure_optimizers()
        tr = self._trainer()
        names = [k for k, _ in self.noise_pred_net.named_parameters()]
        st = ck["optimizer_states"][0]["state"]
        tr.load(2, {k: st[i]["exp_avg"] for i, k in enumerate(names)})
        tr.load(3, {k: st[i]["exp_avg_sq"] for i, k in enumerate(names)}, adam_steps=int(st[0]["step"]))
        if isinstance(ck.get("state_dict", {}).get("ema_model"), Mapping):
            tr.load(4, {k[len("noise_pred_net."):]: v for k, v in ck["state_dict"]["ema_model"].items()})
        for sch in ck.get("lr_schedulers", [])[:1]:
            self.lr_scheduler.last_epoch = int(sch["last_epoch"])
            self.optimizer.param_groups[0]["lr"] = float(sch["_last_lr"][0])
        if "ema" in ck:
            self.ema.optimization_step, self.ema.decay = int(ck["ema"]["optimization_step"]), float(ck["ema"]["decay"])

    # ------------------------------------------------------------------ a5
    def deltas_to_objective(self, deltas, opt_obj, centers=None):
        """generator/diffusion.py:430-471 on an arbitrary deltas tensor (the sampling path itself uses the form fused
        into the trunk kernel; this is for callers that want the objective values)."""
        sign = {'rotate_clockwise': (-1, 0, 0), 'rotate_counterclockwise': (1, 0, 0), 'shift_up': (0, -1, 0), 'shift_down': (0, 1, 0),
                'shift_left': (0, 0, -1), 'shift_right': (0, 0, 1), 'clockwise_up': (-1, -1, 0), 'clockwise_down': (-1, 1, 0),
                'clockwise_left': (-1, 0, -1), 'clockwise_right': (-1, 0, 1), 'counterclockwise_up': (1, -1, 0),
                'counterclockwise_down': (1, 1, 0), 'counterclockwise_left': (1, 0, -1), 'counterclockwise_right': (1, 0, 1)}
        if opt_obj == 'rotate':
            return deltas[..., 0] ** 2
        if opt_obj in sign:
            terms = [deltas[..., j] if c > 0 else -deltas[..., j] for j, c in enumerate(sign[opt_obj]) if c]
            return terms[0] if len(terms) == 1 else terms[0] + terms[1]
        if opt_obj == 'convergence':
            cells, pp = self.grid_size * self.num_pos ** 2, self.num_pos ** 2
            half = (self.grid_size // 2) * pp
            out = []
            for i, c in enumerate(centers):
                c = int(c)
                d = deltas[i * cells:(i + 1) * cells, 0]
                out += [metrics.slicer(d, c * pp - half, c * pp), metrics.slicer(-d, c * pp, c * pp + half)]
            return torch.cat(out, dim=0)
        raise ValueError('opt obj not supported')

    # ------------------------------------------------------------------ a4
    def cond_fn(self, x, t, opt_obj='rotate', object_vertices=None, ori_range=[-1.0, 1.0], convergence_centers=None):
        """d sum(objective(dynamics(x, grid))) / dx, shape of x (generator/diffusion.py:473-504).  In 3-D the FPS starts
        come from the torch CPU generator exactly as the reference's classifier calls would draw them."""
        if self.mode not in ('point', 'point_3d'):
            raise ValueError('model type not supported')
        B = x.shape[0]
        g = self._guidance_for(B, ori_range, object_vertices.reshape(1, *object_vertices.shape[-2:]), 1)
        obj = engine.make_objective(opt_obj, 0)
        tt = int(torch.as_tensor(t).reshape(-1)[0])
        rc = None
        if opt_obj == 'convergence':
            rc = torch.from_numpy(g.rowcoef(torch.as_tensor(convergence_centers))).to(self.device).reshape(1, -1)
        starts = StartStream(g.cfg.num_object_points, g.cfg.sub_batch_size).call(g.rows) if self.mode == 'point_3d' else None
        grad = g.grad(x.reshape(1, B, -1).to(self.device), tt, [obj], rc, starts)
        return grad.reshape(x.shape)

    # ------------------------------------------------------------------ a6
    def get_convergence_centers(self, unguided_sample, object_vertices, batch_size, ori_range=[-1.0, 1.0]):
        g = self._guidance_for(batch_size, ori_range, object_vertices.reshape(1, *object_vertices.shape[-2:]), 1)
        starts = StartStream(g.cfg.num_object_points, g.cfg.sub_batch_size).call(g.sweep_rows) if self.mode == 'point_3d' else None
        return sampler.convergence_centers(g, self.mode, unguided_sample.to(self.device), [0], starts)[0].to(self.device)

    # ------------------------------------------------------------------ harness selection
    def get_best_ids_all_metrics(self, objectives, opt_obj='rotate'):
        """Index of the best gripper for every score of ``opt_obj`` (generator/diffusion.py:391-428)."""
        directions, _ = objective_directions(opt_obj)
        best = {k: (np.argmax if sgn > 0 else np.argmin)([o[k] for o in objectives]) for k, sgn in directions.items()}
        if opt_obj != 'convergence':
            best['success_rate'] = np.argmax([o['success_rate'] for o in objectives])
        return best

    def get_best_ids(self, objectives_unguided, num_grippers, num_objects, opt_obj='rotate'):
        """Per object (scores are laid out object-major, gripper-minor): best flat indices per score (:346-352)."""
        out = []
        for i in range(num_objects):
            block = objectives_unguided[i * num_grippers:(i + 1) * num_grippers]
            out.append({k: v + i * num_grippers for k, v in self.get_best_ids_all_metrics(block, opt_obj=opt_obj).items()})
        return out

    def get_average_best_ids(self, objectives, opt_obj='rotate'):
        """Best gripper by the objective's primary count (:354-389)."""
        directions, primary = objective_directions(opt_obj)
        return (np.argmax if directions[primary] > 0 else np.argmin)([o[primary] for o in objectives])

    # ------------------------------------------------------------------ a1
    def _spec(self, batch: int, ori_range: Sequence[float], npts: int) -> ddist.GuidanceSpec:
        return ddist.GuidanceSpec(self._dyn().handle(), batch, self.grid_size, self.num_pos, ori_range,
                                  self.noise_scheduler.config.num_train_timesteps, npts,
                                  self.sub_batch_size if self.mode == 'point_3d' else 0, self.contraction_dtype)

    def guided_sample(self, batch_idx, batch_size, noise, save_dir, opt_obj='rotate', ori_range=[-1.0, 1.0], unguided_sample=None):
        """All objects' chains of generator/diffusion.py:561-576 as one batch.  Returns (n_objects, B, L, 1).

        Under a process group (torchrun, one rank per GPU) the objects' chains are block-partitioned over the ranks and the
        final samples all-gathered (dgdm_amd/dist.py) - this is what stands in for the reference's nn.DataParallel
        (generator/train.py:86,88); every rank returns all objects' samples, identical to the single-process result."""
        objs = torch.as_tensor(self.object_vertices)
        n = objs.shape[0]
        chains = [(i, opt_obj) for i in range(n)]
        ug = None if unguided_sample is None else unguided_sample.to(self.device)
        if ddist.world_rank()[0] > 1:
            out = ddist.guided_chains_sharded(self._net(), self._spec(batch_size, ori_range, objs.shape[1]), self.noise_scheduler, self.mode,
                                              noise.to(self.device), objs, chains, unguided=ug,
                                              build=lambda o, k: self._guidance_for(batch_size, ori_range, o.detach().cpu(), k))
        else:
            g = self._guidance_for(batch_size, ori_range, objs, n)
            out = sampler.guided_chains(self._net(), g, self.noise_scheduler, self.mode, noise.to(self.device), chains, unguided=ug)
        if ddist.world_rank()[1] == 0:
            tag = f"{opt_obj}_orirange={ori_range[0]:.3f}_{ori_range[1]:.3f}"
            ids = [self.object_ids[i] if self.object_ids is not None else i for i in range(n)]
            self._emit(save_dir, 'vis_guided', tag, out, [str(i) for i in ids])
            if self.simulator is not None and save_dir:                  # :577-619
                arr = out.detach().cpu().numpy()
                sims = [self.simulator(arr[i], [ids[i]], os.path.join(save_dir, 'vis_guided', tag, str(ids[i])), render=self.render_video,
                                       num_cpus=self.num_cpus, num_rot=int((ori_range[1] - ori_range[0]) * 180), ori_range=ori_range) for i in range(n)]
                artefacts.guided_table(self, self._tables(save_dir), sims, opt_obj, ori_range)
        return out

    # ------------------------------------------------------------------ a2
    def guided_sample_multi_object(self, batch_idx, batch_size, noise, save_dir, opt_obj='rotate', ori_range=[-1.0, 1.0]):
        objs = torch.as_tensor(self.object_vertices)
        tag = f"{opt_obj}_orirange={ori_range[0]:.3f}_{ori_range[1]:.3f}"
        rank0 = ddist.world_rank()[1] == 0
        on_step = None
        if save_dir and self.render_plots and rank0:                     # per-step, per-gripper PNGs (:648-674)
            def on_step(i, x):
                for gi, row in enumerate(x.detach().cpu().numpy()):
                    artefacts.plot_fingers(os.path.join(save_dir, 'vis_guided', tag, 'allobj_%d_%d.png' % (batch_idx * batch_size + gi, i)),
                                           row[:, 0], self.mode, self.pts_x_dim, self.pts_z_dim, stacked=False)
        if ddist.world_rank()[0] > 1:       # objects over ranks, gradients all-gathered every step (dist.guided_multi_object_sharded)
            out = ddist.guided_multi_object_sharded(self._net(), self._spec(batch_size, ori_range, objs.shape[1]), self.noise_scheduler, self.mode,
                                                    noise.to(self.device), objs, opt_obj,
                                                    build=lambda o, k: self._guidance_for(batch_size, ori_range, o.detach().cpu(), k), on_step=on_step)
        else:
            g = self._guidance_for(batch_size, ori_range, objs, objs.shape[0])
            out = sampler.guided_multi_object(self._net(), g, self.noise_scheduler, self.mode, noise.to(self.device),
                                              list(range(objs.shape[0])), opt_obj, on_step=on_step)
        if rank0:
            self._emit(save_dir, 'vis_guided', tag, out[None], ["allobj"])
            if self.simulator is not None and save_dir:                  # :675-709: every gripper on all objects
                arr = out.detach().cpu().numpy()
                ids = list(self.object_ids) if self.object_ids is not None else list(range(objs.shape[0]))
                sims = [self.simulator(arr[i:i + 1], ids, os.path.join(save_dir, 'vis_guided', tag, 'allobj_%d' % i), render=self.render_video,
                                       num_cpus=self.num_cpus, num_rot=int((ori_range[1] - ori_range[0]) * 180), ori_range=ori_range)
                        for i in range(arr.shape[0])]
                artefacts.multi_object_table(self, self._tables(save_dir), sims, len(ids), opt_obj, ori_range)
        return out

    def _object_ids(self) -> list:
        return list(self.object_ids) if self.object_ids is not None else list(range(torch.as_tensor(self.object_vertices).shape[0]))

    def _tables(self, save_dir) -> "artefacts.TableLog":
        return artefacts.TableLog(save_dir, self.table_logger)

    def _emit(self, save_dir, sub, tag, samples, names):
        """Where the reference hands `sample.cpu().numpy()` to its simulator (:578-580, :675-683), the samples are kept and saved."""
        arr = samples.detach().cpu().numpy()
        self.last_samples[tag] = arr
        # the geometry the simulator would build from these control values on the host (sim_test_mj.py:254-262 ->
        # assets/finger_sampler.py; sim_test_mj_3d.py:233-237 -> assets/finger_3d.py), decoded on the device
        flat = samples.detach().reshape(-1, samples.shape[-2], 1)
        if samples.is_cuda and (self.mode == 'point_3d') == (flat.shape[1] == 42) and flat.shape[1] >= 8 and flat.shape[1] % 2 == 0:
            geo = engine.finger_decode_3d(flat) if self.mode == 'point_3d' else engine.finger_decode_2d(flat)
            geo = geo.reshape(*samples.shape[:-2], *geo.shape[1:]).cpu().numpy()
        else:
            geo = None
        self.last_geometry[tag] = geo
        if save_dir:
            d = os.path.join(save_dir, sub, tag)
            os.makedirs(d, exist_ok=True)
            for i, nm in enumerate(names):
                np.save(os.path.join(d, f"{nm}.npy"), arr[i])
                if geo is not None:
                    np.save(os.path.join(d, f"{nm}_geometry.npy"), geo[i])

    # ------------------------------------------------------------------ a3 + harness
    def validation_step(self, tensor_data, batch_idx):
        """Loops of generator/diffusion.py:179-339 without plotting / simulation: denoise from the partially noised data,
        unguided chain from noise, then the 12-objective sweep (multi-object chain + per-object chains)."""
        self.sync_model()          # after training steps: the sampling handle reads the trained parameters
        dev = self.device
        data = tensor_data.to(dev)
        B = data.shape[0]
        rs = np.random.RandomState(self.seed)
        noise = torch.from_numpy(rs.randn(B, self.num_points, self.input_dim)).float().to(dev)
        ts = self.num_inference_steps * torch.ones((B,), dtype=torch.int64)
        sample = self.noise_scheduler.add_noise(data, noise, ts)
        net = self._net()
        noise_pred_loss = 0.0
        rank0 = ddist.world_rank()[1] == 0
        plots = bool(self.save_dir) and self.render_plots and rank0
        for i, t in enumerate(self.noise_scheduler.timesteps):
            eps = net.forward(sample, torch.full((B,), int(t), device=dev))
            noise_pred_loss += float(torch.mean((eps - noise) ** 2))
            sample = self.noise_scheduler.step(eps, t, sample).prev_sample
            if plots and batch_idx == 0:                                 # val_vis/<epoch>_<step>.png, sample 0 (:203-231)
                artefacts.plot_fingers(os.path.join(self.save_dir, 'val_vis', '%d_%d.png' % (self.current_epoch, i)),
                                       sample[0, :, 0].detach().cpu().numpy(), self.mode, self.pts_x_dim, self.pts_z_dim, stacked=True)
        stats = {"val/noise pred loss": noise_pred_loss / self.num_inference_steps,
                 "val/denoise loss": float(torch.mean((sample - data) ** 2)),
                 "val/accuracy": float(torch.mean((torch.abs(sample - data) < 0.01).float()))}
        out: Dict[str, Any] = {"stats": stats}
        if batch_idx != 0:
            return out
        imgs: List[str] = []                                             # last-step PNG of every gripper (:273, :292)
        last = len(self.noise_scheduler.timesteps) - 1

        def on_step(i, x, eps):                                          # val_vis_noise/<epoch>_<gripper>_<step>.png (:258-292)
            for gi, row in enumerate(x.detach().cpu().numpy()):
                f = artefacts.plot_fingers(os.path.join(self.save_dir, 'val_vis_noise', '%d_%d_%d.png' % (self.current_epoch, batch_idx * B + gi, i)),
                                           row[:, 0], self.mode, self.pts_x_dim, self.pts_z_dim, stacked=False)
                if i == last:
                    imgs.append(f)
        unguided = sampler.unguided_sample(net, self.noise_scheduler, noise, on_step=on_step if plots else None)
        if rank0:
            self._emit(self.save_dir, 'val_vis_noise', 'unguided', unguided[None], ["unguided"])
        out["unguided"] = unguided
        if self.class_cond:
            if self.object_vertices is None:
                raise ValueError('object vertices not provided')
            sim_unguided, tables = None, self._tables(self.save_dir)
            if rank0 and self.save_dir:
                if self.simulator is not None:                           # one roll-out of the unguided grippers on every object (:301-305)
                    sim_unguided = self.simulator(unguided.detach().cpu().numpy(), self._object_ids(), os.path.join(self.save_dir, 'val_vis_noise'),
                                                  render=self.render_video, num_cpus=self.num_cpus)
                else:
                    tables.skipped("no simulator callable was given to Diffusion(simulator=...): the objective tables of "
                                   "generator/diffusion.py:304-336, 592-619, 697-709 need simulator roll-outs (dynamics/sim_test_mj*.py)")
            with self._draw_ahead(B):
                self._objective_sweep(out, batch_idx, B, noise, unguided, sim_unguided, tables, imgs)
        return out

    def _draw_ahead(self, B: int):
        """The FPS start draws of the whole objective sweep, made ahead of the launches on a worker thread (sampler.StartPlan): they
        depend on nothing the GPU computes, only on the order in which the loops below consume the generator - per objective the
        multi-object chain (per step object after object, :641-643), then the per-object chains (centre sweep, then the steps, :561-576);
        under a process group every rank walks the whole stream and skips what belongs to other ranks (dgdm_amd/dist.py)."""
        import contextlib
        if self.mode != 'point_3d' or self.object_vertices is None:
            return contextlib.nullcontext()
        objs = torch.as_tensor(self.object_vertices)
        n, N, S = objs.shape[0], objs.shape[1], len(self.noise_scheduler.timesteps)
        rows, sweep_rows = B * self.grid_size * self.num_pos ** 2, B * self.grid_size
        world, rank = ddist.world_rank()
        plots = bool(self.save_dir) and self.render_plots and rank == 0
        jobs = []
        for opt_obj in OBJECTIVE_SWEEP:
            if opt_obj != 'convergence':
                if world > 1:
                    mine = ddist.shard_range(n, rank, world)
                    jobs += [(rows, 1, j in mine) for _ in range(S) for j in range(n)]
                elif plots:
                    jobs += [(rows, n, True)] * S
                else:
                    jobs.append((rows, S * n, True))
            mine = ddist.shard_range(n, rank, world)
            for c in range(n):
                if opt_obj == 'convergence':
                    jobs.append((sweep_rows, 1, c in mine))
                jobs.append((rows, S, c in mine))
        return sampler.StartPlan(N, self.sub_batch_size, jobs)

    def _objective_sweep(self, out, batch_idx, B, noise, unguided, sim_unguided, tables, imgs):
        """The 12-objective sweep of generator/diffusion.py:307-339: per objective the multi-object chain, then the per-object chains."""
        for opt_obj in OBJECTIVE_SWEEP:
            rng = [-1.0, 1.0]
            if sim_unguided is not None:
                if not imgs:
                    imgs = [None] * B
                artefacts.unguided_table(self, tables, sim_unguided, imgs, len(self._object_ids()), B, opt_obj, rng, self.mode == 'point_3d')
            if opt_obj != 'convergence':
                out[f"multi/{opt_obj}"] = self.guided_sample_multi_object(batch_idx, B, noise, self.save_dir, opt_obj=opt_obj, ori_range=rng)
            out[f"guided/{opt_obj}"] = self.guided_sample(batch_idx, B, noise, self.save_dir, opt_obj=opt_obj, ori_range=rng,
                                                          unguided_sample=unguided)
