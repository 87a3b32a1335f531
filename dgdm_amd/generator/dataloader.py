"""Finger control-point dataset feeding ``validation_step`` (interface of the reference's generator/dataloader.py:5-20).

Items are the y coordinates of one finger pair's control points, min-max scaled to [-1, 1], shaped (L, 1) float32 - the
tensor layout the sampler and the eps-net work on.  The whole table is normalised once at construction."""
from __future__ import annotations

import numpy as np
from torch.utils.data import Dataset


class GripperDataset(Dataset):
    def __init__(self, gripper_pts, gripper_pts_max_x, gripper_pts_min_x, gripper_pts_max_y, gripper_pts_min_y):
        # the x (and z) columns are fixed grids and are not part of the learned representation; the bounds are kept as
        # attributes because callers de-normalise with them
        self.gripper_pts = gripper_pts
        self.gripper_pts_max_x, self.gripper_pts_min_x = gripper_pts_max_x, gripper_pts_min_x
        self.gripper_pts_max_y, self.gripper_pts_min_y = gripper_pts_max_y, gripper_pts_min_y
        y = np.asarray(gripper_pts)[..., 1].astype(np.float32)
        unit = (y - gripper_pts_min_y) / (gripper_pts_max_y - gripper_pts_min_y)
        self._items = (unit * 2.0 - 1.0)[..., None]              # (n, L, 1)

    def __len__(self) -> int:
        return self._items.shape[0]

    def __getitem__(self, idx):
        return self._items[idx]
