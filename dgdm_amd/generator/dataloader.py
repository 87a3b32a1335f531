"""Finger control-point dataset (reference: generator/dataloader.py:5-20)."""
import numpy as np
from torch.utils.data import Dataset


class GripperDataset(Dataset):
    """Takes the y column of every control point and min-max normalises it to [-1, 1]; items are (L, 1) float32."""

    def __init__(self, gripper_pts, gripper_pts_max_x, gripper_pts_min_x, gripper_pts_max_y, gripper_pts_min_y):
        self.gripper_pts = gripper_pts
        self.gripper_pts_max_x, self.gripper_pts_min_x = gripper_pts_max_x, gripper_pts_min_x
        self.gripper_pts_max_y, self.gripper_pts_min_y = gripper_pts_max_y, gripper_pts_min_y

    def __len__(self):
        return len(self.gripper_pts)

    def __getitem__(self, idx):
        y = self.gripper_pts[idx, :, 1].astype(np.float32)
        span = self.gripper_pts_max_y - self.gripper_pts_min_y
        return ((y - self.gripper_pts_min_y) / span * 2.0 - 1.0).reshape((-1, 1))
