"""Training driver of the 2-D dynamics model (reference: dynamics/main.py:17-208, ``python dynamics/main.py <flags of
dynamics/train_dynamics_2d.sh>``): datasets, the epoch loop around ``Trainer.step``, validation with ``Trainer.inference``,
three-class accuracies, checkpoints (periodic and best), cosine schedule, early stopping.  The compute is ``Trainer`` (HIP);
this file is host glue.  wandb is optional: without it the same scalars go to ``<save_dir>/log.jsonl``."""
from __future__ import annotations

import json
import os
import sys

import torch
from torch.utils.data import DataLoader

from .. import _lib
from .. import dist as _dist
from .dataloader import DynamicsDataset
from .parser import parse
from .trainer import Trainer


def batch_rows(batch, fingers_3d: bool = False):
    """A DataLoader batch [samples, cells, ...] -> the row tensors Trainer.step takes (main.py:25-35): every sample's control ordinates
    and object vertices repeated over its pose cells.  3-D (:29-30, :143-144): control points (rows, 3, L) and clouds (rows, 3, N), built
    as the reference builds them - the batch concatenated `cells` times along dim 0 (row = cell * samples + sample; with the shipped
    --batch_size=1 that is the same order as the scores' (sample, cell))."""
    score = batch['scores']
    cells = score.size(1)
    ori, pos = batch['input_ori'].reshape(-1, 1), batch['input_pos'].reshape(-1, 2)
    if fingers_3d:
        ctrl = torch.cat([batch['ctrlpts'] for _ in range(cells)], 0).moveaxis(-1, -2)
        obj = torch.cat([batch['object_vertices'] for _ in range(cells)], 0).moveaxis(-1, -2)
        return ctrl, score.reshape(-1, 3), ori, pos, obj
    ctrl = batch['ctrlpts'][..., 1].repeat(1, cells).reshape(ori.shape[0], -1)        # y ordinates only (:33)
    obj = batch['object_vertices'].repeat(1, cells, 1).reshape(ori.shape[0], -1)
    return ctrl, score.reshape(-1, 3), ori, pos, obj


def class_accuracy(score: torch.Tensor, pred: torch.Tensor, threshold_std) -> list:
    """Share of rows whose three-way class (below -threshold / between / above threshold) agrees, per output (main.py:37-39)."""
    t = torch.as_tensor(threshold_std, dtype=torch.float32, device=score.device)
    cls = lambda v: (v > t).long() - (v < -t).long()                                    # noqa: E731
    return (cls(score) == cls(pred.to(score.device))).float().mean(dim=0).tolist()


class _Log:
    def __init__(self, args):
        self.path = os.path.join(args.save_dir, 'log.jsonl')
        self.wandb = None
        if getattr(args, 'wandb_id', None):
            try:
                import wandb
                wandb.init(project='dynamics model', config=vars(args), dir=args.save_dir, name=args.wandb_id)
                self.wandb = wandb
            except ImportError:
                pass

    def log(self, scalars):
        if self.wandb is not None:
            self.wandb.log(scalars)
        with open(self.path, 'a') as f:
            f.write(json.dumps(scalars) + '\n')


def validate(args, val_loader, trainer, threshold_std):
    n, loss_sum, acc_sum = 0, 0.0, [0.0, 0.0, 0.0]
    for batch in val_loader:
        ctrl, score, ori, pos, obj = batch_rows(batch, args.fingers_3d)
        pred, loss = trainer.inference(ctrl, score, ori, pos, obj, rows_per_sample=batch['scores'].size(1))
        acc = class_accuracy(score, pred.cpu(), threshold_std)
        loss_sum, acc_sum, n = loss_sum + loss, [a + b for a, b in zip(acc_sum, acc)], n + 1
    n = max(n, 1)
    return (loss_sum / n, *[a / n for a in acc_sum])


def train(args):
    # under a launcher (torchrun --nproc-per-node N): one rank per GPU, data-parallel steps (Trainer._run); every rank loads the same
    # batches in the same order (the CPU generator is synchronised), rank 0 writes the log and the checkpoints
    world, rank, local = _dist.init_from_env()
    if torch.cuda.is_available():
        _lib.device_init(local)
    os.makedirs(args.save_dir, exist_ok=True)
    # the reference reads args.object_mesh_dir (main.py:83,101), which its parser never defines (dynamics/parser.py:22 has --object_dir): that flag it is
    kw = dict(object_max_num_vertices=args.object_max_num_vertices, fingers_3d=args.fingers_3d, object_mesh_dir=getattr(args, 'object_mesh_dir', None) or args.object_dir)
    train_set, val_set = DynamicsDataset(args.data_dir, **kw), DynamicsDataset(args.test_data_dir, **kw)
    threshold_std = train_set.threshold / train_set.std
    train_loader = DataLoader(train_set, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers, drop_last=False)
    val_loader = DataLoader(val_set, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers, drop_last=False)
    trainer = Trainer(args)
    trainer.create_model()
    if args.mode == 'validate':
        if args.checkpoint_path is None:
            raise ValueError('checkpoint path is not specified')
        return validate(args, val_loader, trainer, threshold_std)
    log = _Log(args) if rank == 0 else None
    save = trainer.save_checkpoint if rank == 0 else (lambda path: None)
    best, last_best = float('inf'), 0
    for epoch in range(args.num_epochs):
        loss_sum, acc_sum = 0.0, [0.0, 0.0, 0.0]
        for i, batch in enumerate(train_loader):
            ctrl, score, ori, pos, obj = batch_rows(batch, args.fingers_3d)
            loss, pred = trainer.step(ctrl, score, ori, pos, obj, rows_per_sample=batch['scores'].size(1))
            acc = class_accuracy(score, pred.cpu(), threshold_std)
            loss_sum, acc_sum = loss_sum + loss, [a + b for a, b in zip(acc_sum, acc)]
            rank or log.log({'train/lr': trainer.optimizer.param_groups[0]['lr'], 'train/batch loss': loss, 'train/batch accuracy ori': acc[0],
                     'train/batch accuracy x': acc[1], 'train/batch accuracy y': acc[2]})
            if i % args.save_ckpt_step == 0:
                save(os.path.join(args.save_dir, '%d_%d.pt' % (epoch, i)))
        trainer.lr_scheduler.step()
        nb = max(len(train_loader), 1)
        rank or print('epoch:', epoch, 'loss:', loss_sum / nb, 'accuracy (ori, x, y):', [a / nb for a in acc_sum])
        rank or log.log({'train/average loss': loss_sum / nb, 'train/average accuracy ori': acc_sum[0] / nb, 'train/average accuracy x': acc_sum[1] / nb,
                 'train/average accuracy y': acc_sum[2] / nb})
        if epoch % args.val_step == 0:
            v = validate(args, val_loader, trainer, threshold_std)
            rank or log.log({'val/average loss': v[0], 'val/average accuracy ori': v[1], 'val/average accuracy x': v[2], 'val/average accuracy y': v[3]})
            if v[0] < best:
                best, last_best = v[0], epoch
                save(os.path.join(args.save_dir, 'best.pt'))
            elif epoch - last_best >= args.patience:
                print('early stopping...')
                break
    return trainer


if __name__ == '__main__':
    train(parse(sys.argv[1:]))
