"""Artefact tree of the validation harness: what ``Diffusion.validation_step``,
``guided_sample`` and ``guided_sample_multi_object`` of the reference leave under ``logger.save_dir`` besides the
samples themselves (generator/diffusion.py:203-231, 258-336, 592-619, 648-709).  Host code only.

* per-step scatter plots of the control values, with the reference's file names:
    val_vis/<epoch>_<step>.png                       denoise-from-data loop, sample 0           (:203-231)
    val_vis_noise/<epoch>_<gripper>_<step>.png       unguided chain, every gripper              (:258-292)
    vis_guided/<tag>/allobj_<gripper>_<step>.png     multi-object guided chain, every gripper   (:648-674)
* the tables the reference sends to ``self.logger.log_table`` (wandb).  Without wandb they are written as JSON,
  ``tables/<key with '/' -> '__'>.json`` = {"key", "columns", "data"}; images are referenced by file path, placeholder
  images (the reference's 128x128 white ``wandb.Image``) by null.

The tables hold simulator scores.  The simulator (MuJoCo + Ray: dynamics/sim_test_mj*.py) is outside this package, so it
is a callable the user hands to ``Diffusion(simulator=...)`` with the reference's signature

    simulator(samples [n, L, 1] ndarray, object_ids, save_dir, render=..., num_cpus=..., **kw)
        -> (gripper_imgs, metrics, profiles, profiles_x, profiles_y, finals, videos, save_gripper_dirs)

Without one the plots are still written and the tables are skipped (noted in ``tables/SKIPPED.txt``).
"""
from __future__ import annotations

import json
import os
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from ..dynamics.metrics import metric2objective

ROTATION_FAMILY = ('rotate', 'rotate_clockwise', 'rotate_counterclockwise', 'convergence', 'clockwise_up', 'clockwise_down',
                   'clockwise_left', 'clockwise_right', 'counterclockwise_up', 'counterclockwise_down', 'counterclockwise_left',
                   'counterclockwise_right')


def _plt():
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    return plt


def plot_fingers(path: str, values: np.ndarray, mode: str, pts_x_dim: int = 7, pts_z_dim: int = 3, stacked: bool = True) -> str:
    """One gripper's control values (L,) as the reference draws them: 2-D two stacked axes (left / right finger) over
    linspace(-1, 1, L/2) (:205-214); 3-D either two stacked 3-D axes (:215-231, ``stacked``) or one axis with the fingers
    shifted to y -/+ 1 in orange / green (:274-291, :650-663)."""
    plt = _plt()
    v = np.asarray(values, dtype=np.float64).reshape(-1)
    h = v.shape[0] // 2
    os.makedirs(os.path.dirname(path), exist_ok=True)
    f = plt.figure()
    if mode == 'point':
        xs = np.linspace(-1.0, 1.0, h)
        for k, part in enumerate((v[:h], v[h:])):
            ax = f.add_subplot(211 + k)
            ax.set(xlim=(-1.0, 1.0), ylim=(-1.0, 1.0))
            ax.scatter(xs, part)
    else:
        x_n, z_n = np.meshgrid(np.linspace(-1.0, 1.0, pts_x_dim), np.linspace(-1.0, 1.0, pts_z_dim))
        x_n, z_n = x_n.T.reshape(-1), z_n.T.reshape(-1)
        if stacked:
            for k, part in enumerate((v[:h], v[h:])):
                ax = f.add_subplot(211 + k, projection='3d')
                ax.set(xlim=(-1.0, 1.0), ylim=(-1.0, 1.0), zlim=(-1.0, 1.0))
                ax.scatter(x_n, part, z_n, s=2)
        else:
            ax = f.add_subplot(111, projection='3d')
            ax.set(xlim=(-1.0, 1.0), ylim=(-2.0, 2.0), zlim=(-1.0, 1.0))
            ax.scatter(x_n, v[:h] - 1.0, z_n, s=2, c='orange')
            ax.scatter(x_n, v[h:] + 1.0, z_n, s=2, c='green')
            ax.grid(False)
    f.savefig(path)
    plt.close(f)
    return path


class TableLog:
    """Stand-in for ``self.logger.log_table`` (:322-336, :604-619, :697-709): one JSON file per table key."""

    def __init__(self, save_dir: Optional[str], wandb_logger: Any = None):
        self.dir = os.path.join(save_dir, "tables") if save_dir else None
        self.wandb_logger = wandb_logger
        self.keys: List[str] = []

    @staticmethod
    def _cell(v):
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        if isinstance(v, np.ndarray):
            return v.tolist()
        if isinstance(v, dict):
            return {str(k): TableLog._cell(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [TableLog._cell(x) for x in v]
        return v

    def log_table(self, key: str, columns: Sequence[str], data: Sequence[Sequence[Any]]) -> None:
        self.keys.append(key)
        if self.wandb_logger is not None:
            self.wandb_logger.log_table(key=key, columns=list(columns), data=[list(r) for r in data])
        if self.dir:
            os.makedirs(self.dir, exist_ok=True)
            with open(os.path.join(self.dir, key.replace("/", "__") + ".json"), "w") as f:
                json.dump({"key": key, "columns": list(columns), "data": [[self._cell(c) for c in row] for row in data]}, f, indent=1)

    def skipped(self, why: str) -> None:
        if self.dir:
            os.makedirs(self.dir, exist_ok=True)
            with open(os.path.join(self.dir, "SKIPPED.txt"), "a") as f:
                f.write(why + "\n")


def profile_family(opt_obj: str) -> str:
    """Which of the simulator's profile plots a guided table shows (:584-591)."""
    if opt_obj in ROTATION_FAMILY:
        return 'profiles'
    if opt_obj in ('shift_up', 'shift_down'):
        return 'profiles_x'
    if opt_obj in ('shift_left', 'shift_right'):
        return 'profiles_y'
    raise ValueError('opt obj not supported')


def unguided_table(model, log: TableLog, sim_out, imgs: Sequence[str], num_objects: int, num_grippers: int, opt_obj: str,
                   ori_range: Sequence[float], fingers_3d: bool) -> Dict[str, Any]:
    """The "val/unguided_sample/<opt_obj>_orirange=..." table (:304-336) from one simulator roll-out of the unguided samples."""
    gripper_imgs, metrics, profiles, profiles_x, profiles_y, finals, videos, _ = sim_out
    imgs_all = list(gripper_imgs) if fingers_3d else [imgs[idx] for _ in range(num_objects) for idx in range(len(imgs))]
    lo, hi = int((ori_range[0] + 1) * 180), int((ori_range[1] + 1) * 180)
    sliced = [{k: m[k][lo:hi] for k in m.keys()} for m in metrics]                                          # :304
    objs = [metric2objective(m, opt_obj) for m in sliced]
    keys = list(objs[0].keys())
    average = {k: float(np.mean([o[k] for o in objs])) for k in keys}
    all_best = model.get_best_ids(objs, num_grippers, num_objects, opt_obj=opt_obj)
    best = [{k: objs[b[k]][k] for k in keys} for b in all_best]
    average_best = {k: float(np.mean([b[k] for b in best])) for k in keys}
    per_gripper = [{k: float(np.mean([objs[i * num_grippers + g][k] for i in range(num_objects)])) for k in keys} for g in range(num_grippers)]
    best_avg = int(model.get_average_best_ids(per_gripper, opt_obj=opt_obj))
    rows = [[-1, -1, None, average, None, None, None, None], [-1, -1, None, average_best, None, None, None, None],
            [-1, best_avg, imgs[best_avg] if best_avg < len(imgs) else None, per_gripper[best_avg], None, None, None, None]]
    rows += [[i // num_grippers, i % num_grippers, g, o, p, px, py, fi]
             for i, (g, o, p, px, py, fi) in enumerate(zip(imgs_all, objs, profiles, profiles_x, profiles_y, finals))]
    log.log_table("val/unguided_sample/%s_orirange=%.3f_%.3f" % (opt_obj, ori_range[0], ori_range[1]),
                  ["object_idx", "gripper_idx", "gripper", "objective", "profile", "profile_x", "profile_y", "final"], rows)
    return {"average": average, "average_best": average_best, "best_average_gripper": best_avg}


def guided_table(model, log: TableLog, per_object_sim: Sequence[Any], opt_obj: str, ori_range: Sequence[float]) -> Optional[Dict[str, float]]:
    """The "val/guided_sample/<opt_obj>_orirange=..." table (:577-619): per object the best gripper for every score."""
    fam = profile_family(opt_obj)
    all_imgs, all_obj, all_prof, all_fin, all_vid, all_dirs = [], [], [], [], [], []
    for sim_out in per_object_sim:
        gripper_imgs, metrics, profiles, profiles_x, profiles_y, finals, videos, dirs = sim_out
        if len(metrics) == 0:
            continue
        objs = [metric2objective(m, opt_obj) for m in metrics]
        prof = {'profiles': profiles, 'profiles_x': profiles_x, 'profiles_y': profiles_y}[fam]
        best = model.get_best_ids_all_metrics(objs, opt_obj=opt_obj)
        pick = lambda seq: {k: seq[best[k]] for k in best}                                                   # noqa: E731
        all_obj.append(pick(objs)); all_imgs.append(pick(gripper_imgs)); all_prof.append(pick(prof))
        all_fin.append(pick(finals)); all_vid.append(pick(videos)); all_dirs.append(pick(dirs))
    if not all_obj:
        return None
    average_best = {k: float(np.mean([o[k][k] for o in all_obj])) for k in all_obj[0].keys()}
    rows = [[-1, None, average_best, None, None, [None], ""]]
    rows += [[i, all_imgs[i][k], all_obj[i][k], all_prof[i][k], all_fin[i][k], list(all_vid[i][k]), all_dirs[i][k]]
             for i in range(len(all_obj)) for k in all_obj[i].keys()]
    log.log_table("val/guided_sample/%s_orirange=%.3f_%.3f" % (opt_obj, ori_range[0], ori_range[1]),
                  ["object_idx", "gripper", "objective", "profile", "final", "last_img", "gripper_dir"], rows)
    return average_best


def multi_object_table(model, log: TableLog, per_gripper_sim: Sequence[Any], num_objects: int, opt_obj: str,
                       ori_range: Sequence[float]) -> Optional[Dict[str, Any]]:
    """The "val/guided_sample/allobj_<opt_obj>_orirange=..." table (:675-709): every gripper simulated on all objects,
    scores averaged over the objects, best gripper per score."""
    all_obj, all_dirs, all_imgs, all_vid = [], [], [], []
    for sim_out in per_gripper_sim:
        gripper_imgs, metrics, _, _, _, _, videos, dirs = sim_out
        if len(metrics) != num_objects:
            continue
        objs = [metric2objective(m, opt_obj) for m in metrics]
        all_obj.append({k: float(np.mean([o[k] for o in objs])) for k in objs[0].keys()})
        all_dirs.append(dirs[0]); all_imgs.append(gripper_imgs[0]); all_vid.append(sum((list(v) for v in videos), []))
    if not all_obj:
        return None
    best = model.get_best_ids_all_metrics(all_obj, opt_obj=opt_obj)
    rows = [[all_imgs[best[k]], all_obj[best[k]], all_vid[best[k]], all_dirs[best[k]]] for k in best]
    log.log_table("val/guided_sample/allobj_%s_orirange=%.3f_%.3f" % (opt_obj, ori_range[0], ori_range[1]),
                  ["gripper", "objective", "last_img", "gripper_dir"], rows)
    return {k: all_obj[best[k]] for k in best}
