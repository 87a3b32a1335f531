"""Command-line flags shared by every entry point (reference: dynamics/parser.py:3-41).

Same names, types, defaults and help strings, so the shipped ``guided_sample_{2d,3d}.sh`` lines parse unchanged."""
import argparse

# (flag, type | 'flag', default, help)
_FLAGS = [
    ("batch_size", int, 1024, None),
    ("use_sub_batch", "flag", None, "use sub batch to avoid OOM"),
    ("sub_bs", int, 1024, "sub batch size for training"),
    ("num_epochs", int, 1000, "number of epochs for training"),
    ("num_fingers", int, 1000, "number of fingers"),
    ("ctrlpts_dim", int, 14, None),
    ("ctrlpts_x_dim", int, 7, None),
    ("ctrlpts_z_dim", int, 3, None),
    ("learning_rate", float, 1e-4, "learning rate for optimizer"),
    ("lr_warmup_steps", int, 100, "learning rate warmup steps for optimizer"),
    ("weight_decay", float, 0, "weight decay for optimizer"),
    ("patience", int, 500, "patience for early stopping when training dynamics model"),
    ("checkpoint_path", str, None, "path to load dynamics model checkpoints"),
    ("save_dir", str, None, "path to save model checkpoints"),
    ("wandb_id", str, None, "wandb id"),
    ("data_dir", str, "", "path to data directory"),
    ("test_data_dir", str, "", "path to test data directory"),
    ("object_dir", str, "", "path to object directory"),
    ("num_workers", int, 4, "number of workers for dataloader"),
    ("mode", str, "train", "train or test"),
    ("grid_size", int, 360, "number of initial orientations sampled for each object"),
    ("num_pos", int, 9, "number of initial positions sampled for each object"),
    ("save_ckpt_step", int, 10, "step to save model checkpoints"),
    ("val_step", int, 100, "step to validate model"),
    ("num_train_timesteps", int, 1000, "number of training timesteps for diffusion model"),
    ("num_timesteps_per_batch", int, 1, "number of timesteps per batch"),
    ("num_inference_steps", int, 100, "number of inference steps for diffusion model"),
    ("ema_power", float, 0.75, "ema power"),
    ("object_max_num_vertices", int, 10, "max number of vertices for object encoder"),
    ("diffusion_checkpoint_path", str, None, "path to load diffusion model checkpoints"),
    ("classifier_guidance", "flag", None, "use classifier guidance"),
    ("num_cpus", int, 4, "number of cpus used in parallel for simulation"),
    ("fingers_3d", "flag", None, "use 3d fingers"),
    ("render_video", "flag", None, "render videos visualizing interactions of fingers and objects"),
    ("seed", int, 0, "random seed"),
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
