"""Entry point behind ``python generator/train.py --mode=test ...`` (reference: generator/train.py:38-166).

Keeps the reference's flags (dynamics/parser.py) and construction order: synthetic finger set from ``RandomState(idx)``
(:43-58), U-Net + DDIM scheduler (:80-83), frozen dynamics model loaded from ``--checkpoint_path`` (:84-92), normalised
object point sets (:93-124), ``Diffusion`` (:129), then - in test mode - one pass of ``validation_step`` per batch of
fingers, which is what Lightning's ``trainer.validate`` does (:152-156).  ``--mode=train`` (the flags of
``generator/train_diffusion_{2d,3d}.sh``) is ``trainer.fit`` (:158-162): epochs over the shuffled 90 % split, one
``training_step`` + ``on_train_batch_end`` per batch on the GPU (csrc/unet_train.hip), the cosine schedule stepped per epoch,
``validation_step`` on the held-out 10 % every ``--val_step`` epochs, a Lightning-shaped checkpoint per epoch (the last ten and
``last.ckpt``, as the ModelCheckpoint of :137-146 keeps them).

Assets the image does not have are substituted, loudly:
* no checkpoint files  -> deterministic random-init weights (dgdm_amd.synth);
* no Icons-50 / scanned meshes (and no open3d, cv2) -> synthetic contours / surface-sampled clouds with the same
  normalisation; an ``--object_dir`` holding ``objects.npy`` ([n, vertices, 2|3], metres) is used when present.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from .. import synth
from ..dynamics.parser import parse
from ..dynamics.profile_forward_2d import ProfileForward2DModel
from ..dynamics.profile_forward_3d import ProfileForward3DModel
from ..scheduler import DDIMScheduler
from .dataloader import GripperDataset
from .diffusion import Diffusion
from .diffusion_utils import ConditionalUnet1D

OBJECT_IDS = [10000, 2009, 2114, 2082, 1041, 2048, 1045, 1019]     # Icons-50 test ids, generator/train.py:36
OBJECT_NAMES_3D = ["3D_Dollhouse_Swing", "BABY_CAR", "Ecoforms_Plant_Container_B4_Har", "Threshold_Bamboo_Ceramic_Soap_Dish",
                   "Squirt_Strain_Fruit_Basket",
                   "Office_Depot_Canon_CLI_8CMY_Remanufactured_Ink_Cartridges_Color_Cyan_Magenta_Yellow_3_count"]   # assets/object_names_test.txt


def finger_control_points(num_fingers: int, fingers_3d: bool) -> np.ndarray:
    """generator/train.py:43-58 and assets/finger_3d.py:82-88: per finger idx, RandomState(idx) draws the y of the left then
    the right control points; x (and z) are fixed grids.  Shape (n, 14, 2) or (n, 42, 3)."""
    out = []
    for idx in range(num_fingers):
        rs = np.random.RandomState(idx)
        if fingers_3d:
            yl, yr = rs.uniform(-0.1, 0, size=21), rs.uniform(-0.1, 0, size=21)
            xg, zg = np.meshgrid(np.linspace(-0.12, 0.12, 7), np.linspace(0, 0.12, 3))
            side = lambda y: np.stack([xg.T.reshape(-1), y, zg.T.reshape(-1)], axis=-1)      # noqa: E731
            out.append(np.concatenate((side(yl), side(yr)), axis=0))
        else:
            x = np.linspace(-0.12, 0.12, 7)
            yl, yr = rs.uniform(-0.045, 0.015, size=7), rs.uniform(-0.045, 0.015, size=7)
            out.append(np.concatenate((np.stack([x, yl], axis=-1), np.stack([x, yr], axis=-1)), axis=0))
    return np.stack(out, axis=0)


def _load_or_synth(path, spec, seed, what):
    if path and os.path.exists(path):
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        return {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    print(f"[dgdm_amd] {what}: '{path}' not found - using deterministic random-init weights (seed {seed})", file=sys.stderr)
    return synth.synth_state_dict(spec, seed)


def _objects(args, fingers_3d: bool):
    f = os.path.join(args.object_dir or "", "objects.npy")
    nv = args.object_max_num_vertices
    if os.path.isfile(f):
        raw = torch.from_numpy(np.load(f)).float()
        lo, hi = (torch.tensor([-0.1, -0.1, 0.0]), torch.tensor([0.1, 0.1, 0.12])) if fingers_3d else (torch.tensor([-0.05] * 2), torch.tensor([0.05] * 2))
        return (raw - lo) / (hi - lo) * 2.0 - 1.0, list(range(raw.shape[0]))            # generator/train.py:94-124
    print("[dgdm_amd] no objects.npy under --object_dir - using synthetic objects", file=sys.stderr)
    if fingers_3d:
        return torch.stack([synth.synth_object_3d(i, nv) for i in range(len(OBJECT_NAMES_3D))]), list(OBJECT_NAMES_3D)
    return torch.stack([synth.synth_object_2d(i, nv) for i in range(len(OBJECT_IDS))]), list(OBJECT_IDS)


def fit(model: Diffusion, pts: np.ndarray, args, bounds, dev) -> Diffusion:
    """``trainer.fit(diffusion_model, train_loader, val_loader)`` (generator/train.py:44-45, 66-67, 147-162) without Lightning: the
    loop its ``LightningTrainer(max_epochs=num_epochs, check_val_every_n_epoch=val_step)`` runs.  Under torchrun every rank takes the
    slice ``indices[rank::world]`` of the epoch's permutation (DistributedSampler) and the gradients are averaged (DDP)."""
    from torch.utils.data import DataLoader
    from .. import dist as ddist
    n = pts.shape[0]
    train_ids, val_ids = list(range(int(n * 0.9))), list(range(int(n * 0.9), n))
    train_set, val_set = GripperDataset(pts[train_ids], *bounds), GripperDataset(pts[val_ids], *bounds)
    world, rank = ddist.world_rank()
    if not hasattr(model, "lr_scheduler"):      # a resumed model (load_checkpoint) carries its schedule position
        model.configure_optimizers()
    model.to(dev)
    if world > 1:
        # Lightning wraps the module in DistributedDataParallel, whose constructor broadcasts rank 0's parameters and buffers (generator/
        # train.py:147-162 with strategy 'ddp'): without it every rank would start from its own random eps-net and the averaged gradients
        # would be taken at different weights.  A resumed model (same checkpoint file on every rank) already agrees and keeps its handle.
        if getattr(model, "_unet_trainer", None) is None:
            with torch.no_grad():
                for t in list(model.noise_pred_net.parameters()) + list(model.noise_pred_net.buffers()):
                    t.copy_(ddist.broadcast_from_rank0(t.detach().clone()))
        # the noise / timestep draws of get_stats come from the device generator: a different stream per rank, as DDP workers have
        torch.cuda.manual_seed((int(torch.initial_seed()) + 7919 * rank) & 0x7FFFFFFFFFFFFFFF)
    ckdir = os.path.join(args.save_dir, "checkpoints") if args.save_dir else None
    if ckdir and rank == 0:
        os.makedirs(ckdir, exist_ok=True)
    step, log = int(getattr(model, "global_step", 0)), []
    # save_top_k = 10 also counts the epoch files a resumed run finds in place
    kept = sorted(os.path.join(ckdir, f) for f in os.listdir(ckdir) if f.startswith("epoch=") and f.endswith(".ckpt")) if ckdir and os.path.isdir(ckdir) else []
    val_loader = DataLoader(val_set, batch_size=args.batch_size, shuffle=False, num_workers=0, drop_last=False)

    def validate(limit=None):
        # Every rank walks the WHOLE validation set, batch by batch, and that is deliberate: validation_step's guided chains are sharded
        # over the ranks inside each batch (objectives / objects over ranks, results all-gathered: diffusion.py guided_* ->
        # dist.guided_chains_sharded), so all ranks must enter the same batches.  A DistributedSampler on top (the reference's Lightning
        # default) would hand the ranks different batches under those collectives.
        model.eval()
        out = []
        with torch.no_grad():
            for bi, batch in enumerate(val_loader):
                if limit is not None and bi >= limit:
                    break
                out.append(model.validation_step(batch, bi)["stats"])
        model.train()
        return out
    if len(val_set):
        validate(limit=2)                      # Lightning's sanity check (num_sanity_val_steps = 2) before the first epoch
    for epoch in range(model.lr_scheduler.last_epoch, args.num_epochs):
        model.current_epoch = epoch
        if world == 1:
            loader = DataLoader(train_set, batch_size=args.batch_size, shuffle=True, num_workers=0, drop_last=False)
        else:                                   # DistributedSampler(shuffle=True, seed=0): randperm from Generator(seed + epoch), padded, strided
            gen = torch.Generator()
            gen.manual_seed(epoch)
            idx = torch.randperm(len(train_set), generator=gen).tolist()
            total = -(-len(idx) // world) * world
            idx = (idx + idx[:total - len(idx)])[rank:total:world]
            loader = DataLoader(torch.utils.data.Subset(train_set, idx), batch_size=args.batch_size, shuffle=False, num_workers=0, drop_last=False)
        losses = []
        for bi, batch in enumerate(loader):
            losses.append(float(model.training_step(batch, bi)))
            model.on_train_batch_end(None, batch, bi)
            step += 1
        model.lr_scheduler.step()
        rec = {"epoch": epoch, "train/loss": float(np.mean(losses)), "lr": model.lr_scheduler.get_last_lr()[0], "ema_decay": model.ema.decay}
        if (epoch + 1) % max(1, args.val_step) == 0 and len(val_set):
            vs = validate()
            rec.update({k: float(np.mean([v[k] for v in vs])) for k in vs[0]})
        log.append(rec)
        if rank == 0:
            print("[dgdm_amd] " + ", ".join(f"{k}={v:.6g}" if isinstance(v, float) else f"{k}={v}" for k, v in rec.items()), flush=True)
            if ckdir:
                ck = model.checkpoint(epoch=epoch + 1, global_step=step)
                f = os.path.join(ckdir, "epoch=%04d.ckpt" % epoch)
                torch.save(ck, f)
                torch.save(ck, os.path.join(ckdir, "last.ckpt"))
                if f in kept:                  # a run resumed from an older checkpoint re-saves epochs that are already on the list
                    kept.remove(f)
                kept.append(f)
                while len(kept) > 10:          # save_top_k = 10, monitor = 'epoch', mode = 'max'
                    old = kept.pop(0)
                    if os.path.exists(old):
                        os.remove(old)
    model.sync_model()
    model.train_log = log
    if world > 1:
        # every rank took the same Adam steps on the same averaged gradients from the same start: the replicas must be bit-identical
        flat = torch.cat([p.detach().reshape(-1).double() for p in model.noise_pred_net.parameters()])
        sums = ddist.all_gather_rows(torch.stack([flat.sum(), flat.abs().sum(), (flat * torch.arange(1, flat.numel() + 1, device=flat.device, dtype=flat.dtype)).sum()]))
        if not bool((sums == sums[0]).all()):
            raise RuntimeError(f"data-parallel replicas of the eps-net diverged: per-rank parameter checksums {sums.tolist()}")
        if rank == 0:
            print(f"[dgdm_amd] {world} ranks hold identical eps-net parameters (checksum {float(sums[0][1]):.9g})", flush=True)
    return model


def train(args):
    dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else int(os.environ.get("LOCAL_RANK", "0")))
    pts = finger_control_points(args.num_fingers, args.fingers_3d)
    max_y, min_y = (0.0, -0.1) if args.fingers_3d else (0.015, -0.045)
    dataset = GripperDataset(pts, 0.12, -0.12, max_y, min_y)
    L = args.ctrlpts_dim
    unet = ConditionalUnet1D(input_dim=1, global_cond_dim=0, down_dims=[128, 256], diffusion_step_embed_dim=32)
    mode = 'point_3d' if args.fingers_3d else 'point'
    scheduler = DDIMScheduler(num_train_timesteps=args.num_train_timesteps, beta_schedule='squaredcos_cap_v2', clip_sample=True,
                              prediction_type='epsilon')
    classifier, objects, object_ids = None, None, None
    if args.classifier_guidance:
        if args.fingers_3d:
            classifier = ProfileForward3DModel(output_ch=3, params_ch=L)
            spec = synth.dyn3d_spec(L)
        else:
            classifier = ProfileForward2DModel(output_ch=3, params_ch=L, object_ch=2 * args.object_max_num_vertices)
            spec = synth.dyn2d_spec(L, 2 * args.object_max_num_vertices)
        classifier.load_state_dict(_load_or_synth(args.checkpoint_path, spec, 22 + int(args.fingers_3d), "dynamics checkpoint"))
        classifier.eval().requires_grad_(False).to(dev)
        objects, object_ids = _objects(args, args.fingers_3d)
    model = Diffusion(noise_pred_net=unet, noise_scheduler=scheduler, num_inference_steps=args.num_inference_steps, mode=mode, input_dim=1,
                      num_points=L, learning_rate=args.learning_rate, lr_warmup_steps=args.lr_warmup_steps, ema_power=args.ema_power,
                      class_cond=args.classifier_guidance, classifier_model=classifier, grid_size=args.grid_size, num_pos=args.num_pos,
                      object_vertices=objects, object_ids=object_ids, num_cpus=args.num_cpus, pts_x_dim=args.ctrlpts_x_dim,
                      pts_z_dim=args.ctrlpts_z_dim, sub_batch_size=args.sub_bs, render_video=args.render_video, seed=args.seed)
    if args.mode != 'test':                 # generator/train.py:158-162
        if args.diffusion_checkpoint_path is not None:
            print('loading diffusion checkpoint from', args.diffusion_checkpoint_path)
            model.to(dev)
            model.load_checkpoint(torch.load(args.diffusion_checkpoint_path, map_location="cpu", weights_only=False))
        model.save_dir = args.save_dir or None
        return fit(model, pts, args, (0.12, -0.12, max_y, min_y), dev), []
    ck = _load_or_synth(args.diffusion_checkpoint_path, [("ema_nets.noise_pred_net." + k, s) for k, s in synth.unet_spec()], 11,
                        "diffusion checkpoint")
    res = model.load_state_dict(ck)
    lost = [k for k in res.missing_keys if k.startswith("ema_nets.noise_pred_net.")]
    if lost or res.unexpected_keys:
        raise KeyError(f"diffusion checkpoint does not match ConditionalUnet1D: missing {lost[:5]}{'...' if len(lost) > 5 else ''}, "
                       f"unexpected {list(res.unexpected_keys)[:5]}")
    model.eval().to(dev)
    model.save_dir = args.save_dir or None
    if model.save_dir:
        os.makedirs(model.save_dir, exist_ok=True)
    results = []
    n_batches = len(dataset) // args.batch_size                       # DataLoader(drop_last=True), :69
    with torch.no_grad():
        for bi in range(n_batches):
            batch = torch.from_numpy(np.stack([dataset[i] for i in range(bi * args.batch_size, (bi + 1) * args.batch_size)]))
            res = model.validation_step(batch, bi)
            if int(os.environ.get("RANK", "0")) == 0:
                print(f"[dgdm_amd] batch {bi}: " + ", ".join(f"{k}={v:.5f}" for k, v in res["stats"].items()))
            results.append(res)
    return model, results


def main(argv=None):
    """Single process: one GPU.  Under torchrun (`torchrun --nproc-per-node N generator/train.py ...`): one rank per GPU, the
    guided chains sharded over the ranks (dgdm_amd/dist.py) where the reference wraps the classifier in nn.DataParallel (:86,88)."""
    from .. import _lib
    from .. import dist as ddist
    world, rank, local = ddist.init_from_env()
    _lib.device_init(local)
    try:
        train(parse(argv))
    finally:
        if world > 1:
            import torch.distributed as td
            td.destroy_process_group()


if __name__ == "__main__":
    main()
