"""Command-line flags shared by every entry point (reference: dynamics/parser.py:3-41).

Same flag names, types and defaults, so the shipped ``guided_sample_{2d,3d}.sh`` command lines parse unchanged."""
import argparse

# (flag, type | 'flag', default, help)
_FLAGS = [
    ("batch_size", int, 1024, "fingers per batch"),
    ("use_sub_batch", "flag", None, "split large classifier calls"),
    ("sub_bs", int, 1024, "rows per classifier call in 3-D guidance (fixes the FPS start partition)"),
    ("num_epochs", int, 1000, "training epochs (training is not part of this package)"),
    ("num_fingers", int, 1000, "size of the synthetic finger set"),
    ("ctrlpts_dim", int, 14, "control points per finger pair: 14 (2-D) or 42 (3-D)"),
    ("ctrlpts_x_dim", int, 7, "control net size along the finger"),
    ("ctrlpts_z_dim", int, 3, "control net size across the finger (3-D)"),
    ("learning_rate", float, 1e-4, "optimizer step size"),
    ("lr_warmup_steps", int, 100, "warm-up steps"),
    ("weight_decay", float, 0, "L2 penalty"),
    ("patience", int, 500, "early-stopping patience of dynamics training"),
    ("checkpoint_path", str, None, "dynamics model checkpoint (.pt, DataParallel state_dict)"),
    ("save_dir", str, None, "output directory"),
    ("wandb_id", str, None, "wandb run id"),
    ("data_dir", str, "", "training data"),
    ("test_data_dir", str, "", "held-out data"),
    ("object_dir", str, "", "objects: Icons-50 .npy (2-D) / scanned meshes (3-D) / objects.npy"),
    ("num_workers", int, 4, "DataLoader workers"),
    ("mode", str, "train", "'train' or 'test' (guided sampling)"),
    ("grid_size", int, 360, "orientations in the guidance grid"),
    ("num_pos", int, 9, "positions per axis in the guidance grid"),
    ("save_ckpt_step", int, 10, "checkpoint period"),
    ("val_step", int, 100, "validation period"),
    ("num_train_timesteps", int, 1000, "diffusion timesteps T"),
    ("num_timesteps_per_batch", int, 1, "timesteps drawn per training batch"),
    ("num_inference_steps", int, 100, "denoise steps S"),
    ("ema_power", float, 0.75, "EMA decay exponent"),
    ("object_max_num_vertices", int, 10, "points per object"),
    ("diffusion_checkpoint_path", str, None, "diffusion checkpoint (.ckpt, Lightning)"),
    ("classifier_guidance", "flag", None, "guide sampling with the dynamics model"),
    ("num_cpus", int, 4, "CPU workers of the (external) simulator"),
    ("fingers_3d", "flag", None, "3-D fingers / PointNet++ dynamics model"),
    ("render_video", "flag", None, "simulator videos (external)"),
    ("seed", int, 0, "seed of the start noise"),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    for name, kind, default, helptext in _FLAGS:
        if kind == "flag":
            p.add_argument("--" + name, action="store_true", help=helptext)
        else:
            p.add_argument("--" + name, type=kind, default=default, help=helptext)
    return p


def parse(argv=None):
    return build_parser().parse_args(argv)
