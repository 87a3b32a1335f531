"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

from dgdm_amd import synth
from oracle import dgdm_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def unpack_starts(flat, lens):
    """Inverse of make_golden.RandintSpy.packed: list of per-call index tensors."""
    out, o = [], 0
    for n in lens:
        out.append(torch.from_numpy(np.asarray(flat[o:o + int(n)], dtype=np.int64)))
        o += int(n)
    return out


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def unet_sd(seed):
    return synth.synth_state_dict(synth.unet_spec(), int(seed))


def dyn2d_sd(seed, nv):
    return synth.synth_state_dict(synth.dyn2d_spec(14, 2 * int(nv)), int(seed))


def dyn3d_sd(seed):
    return synth.synth_state_dict(synth.dyn3d_spec(42), int(seed))


def setup(mode, unet, dyn, T, S, L, G, P, sub=1024):
    sch = orc.DDIM(int(T))
    sch.set_timesteps(int(S))
    return orc.Setup(mode, unet, dyn, sch, int(L), int(G), int(P), int(sub))


def finger_err(a, b):
    """Per-finger L2 distance between two sample batches (..., L, 1)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return (a - b).reshape(-1, a.shape[-2] * a.shape[-1]).norm(dim=1)


def oracle_band(fn):
    """Runs an oracle computation with 1 and with the default number of CPU threads.

    PyTorch's CPU kernels sum in a different order for different thread counts, and the gradient of a ReLU network is
    discontinuous where a pre-activation crosses zero - so the reference implementation does not reproduce ITSELF to 1e-4
    on every finger of a free-running chain (measured: 5.6e-4 on one finger of a 2-D 'shift_up' chain, tests/test_gpu_api.py).
    Free-running chains are therefore accepted when every finger is within tolerance of at least one of the two runs."""
    n = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        a = fn()
    finally:
        torch.set_num_threads(n)
    return [a, fn()]


def within_band(out, refs, tol):
    err = torch.stack([finger_err(out, r) for r in refs]).min(dim=0).values
    return bool((err < tol).all()), err


def train2d_data(seed, n_grippers=3, n_poses=128, L=14, nv=100):
    """A synthetic training batch shaped like dynamics/main.py:132-145 hands it to Trainer.step: every (gripper, object) sample
    repeated over its pose grid, scores per row.  Inputs only."""
    rs = np.random.RandomState(seed)
    rep = lambda a: np.repeat(a, n_poses, axis=0)                                   # noqa: E731
    ctrl = rep(rs.uniform(-1, 1, (n_grippers, L))).astype(np.float32)
    obj = rep(rs.uniform(-1, 1, (n_grippers, 2 * nv))).astype(np.float32)
    ori = rs.uniform(-1, 1, (n_grippers * n_poses, 1)).astype(np.float32)
    pos = rs.uniform(-1, 1, (n_grippers * n_poses, 2)).astype(np.float32)
    score = rs.normal(0, 1, (n_grippers * n_poses, 3)).astype(np.float32)
    return [torch.from_numpy(a) for a in (ctrl, score, ori, pos, obj)]


def train3d_data(seed, n_grippers=2, n_poses=4, L=42, N=512):
    """A synthetic 3-D training batch shaped like dynamics/main.py hands it to Trainer.step with --fingers_3d: every (gripper, object)
    sample repeated over its pose rows - control points (rows, 3, L), scores (rows, 3), orientation (rows, 1), position (rows, 2), cloud
    (rows, 3, N).  Inputs only."""
    from dgdm_amd import synth
    rs = np.random.RandomState(seed)
    rep = lambda a: np.repeat(a, n_poses, axis=0)                                   # noqa: E731
    ctrl = rep(rs.uniform(-1, 1, (n_grippers, 3, L))).astype(np.float32)
    obj = rep(np.stack([synth.synth_object_3d(70 + seed + i, N).numpy().T for i in range(n_grippers)])).astype(np.float32)
    ori = rs.uniform(-1, 1, (n_grippers * n_poses, 1)).astype(np.float32)
    pos = rs.uniform(-1, 1, (n_grippers * n_poses, 2)).astype(np.float32)
    score = rs.normal(0, 1, (n_grippers * n_poses, 3)).astype(np.float32)
    return [torch.from_numpy(np.ascontiguousarray(a)) for a in (ctrl, score, ori, pos, obj)]


def sample_idx(name, numel, k=96):
    """Which entries of a parameter tensor the training fixture keeps (all of a small one)."""
    import zlib
    if numel <= k:
        return np.arange(numel)
    return np.sort(np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff).choice(numel, k, replace=False))



def write_synth_dataset(root, seed, n_files=3, cells=12, n_ctrl=14, n_verts=(5, 7, 4)):
    """Synthetic dynamics-training files in the simulator's format (dynamics/dataloader.py:41-55 reads them): inputs only."""
    rs = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    for i in range(n_files):
        d = {"ctrlpts": np.stack([np.linspace(-0.12, 0.12, n_ctrl), rs.uniform(-0.045, 0.015, n_ctrl)], 1),
             "delta_theta": rs.normal(0, 0.05, cells), "delta_pos": rs.normal(0, 0.003, (cells, 2)),
             "obj_theta": rs.uniform(0, 2 * np.pi, cells), "obj_pos": rs.uniform(-0.03, 0.03, (cells, 3)),
             "object_vertices": rs.uniform(-0.05, 0.05, (n_verts[i % len(n_verts)], 2))}
        np.savez(os.path.join(root, "sample_%02d.npz" % i), d)

