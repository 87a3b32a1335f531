"""``DynamicsDataset`` of the dynamics-model training driver (reference: dynamics/dataloader.py:7-78): one ``.npz`` per simulated
(gripper, object) pair holding the pose grid it was rolled out on.  Host code; the 2-D branch (the 3-D one samples object points
from meshes with trimesh, which the 3-D training path - not built - would need).

File format, as the reference's simulator writes it (``np.savez(path, dict)`` -> key ``arr_0``, a pickled dict):
    ctrlpts [n, 2] metres, delta_theta [cells], delta_pos [cells, 2], obj_theta [cells] in [0, 2 pi), obj_pos [cells, >=2] metres,
    object_vertices [v <= object_max_num_vertices, 2] metres.
"""
from __future__ import annotations

import os
from typing import Dict, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

# score normalisation and class thresholds (dataloader.py:11-16); index 0 = 3-D, 1 = 2-D
SCORE_STD = (np.array([0.0312, 0.0016, 0.0026]), np.array([0.0565, 0.0026, 0.0047]))
SCORE_THRESHOLD = (np.array([0.02, 0.001, 0.001]), np.array([0.03, 0.002, 0.003]))
# workspace boxes of dynamics/main.py:61-82 (2-D): metres -> [-1, 1]
GRIPPER_BOX_2D = ((-0.12, 0.12), (-0.045, 0.015))
OBJECT_BOX_2D = ((-0.05, 0.05), (-0.05, 0.05))


def _to_unit(a: np.ndarray, box: Sequence[Sequence[float]]) -> np.ndarray:
    out = np.array(a, dtype=np.float64, copy=True)
    for axis, (lo, hi) in enumerate(box):
        out[..., axis] = (out[..., axis] - lo) / (hi - lo) * 2.0 - 1.0
    return out


class DynamicsDataset(Dataset):
    def __init__(self, dataset_dir: str, object_max_num_vertices: int = 10, fingers_3d: bool = False, gripper_box=GRIPPER_BOX_2D,
                 object_box=OBJECT_BOX_2D, **unused):
        if fingers_3d:
            raise NotImplementedError("the 3-D dynamics dataset (mesh point sampling) belongs to the 3-D training path, which is not built")
        self.std, self.threshold = SCORE_STD[1], SCORE_THRESHOLD[1]
        self.gripper_box, self.object_box, self.object_max_num_vertices = gripper_box, object_box, object_max_num_vertices
        self.data_files = sorted(os.path.join(root, f) for root, _, files in os.walk(dataset_dir) for f in files if f.endswith('.npz'))

    def __len__(self) -> int:
        return len(self.data_files)

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        d = np.load(self.data_files[idx], allow_pickle=True)['arr_0'].item()
        scores = np.stack([d['delta_theta'] / self.std[0], d['delta_pos'][:, 0] / self.std[1], d['delta_pos'][:, 1] / self.std[2]], axis=1)
        verts = torch.from_numpy(_to_unit(d['object_vertices'], self.object_box)).float()
        verts = torch.cat([verts, torch.zeros(self.object_max_num_vertices - verts.shape[0], 2)], dim=0)       # zero-padded (:73)
        return {'ctrlpts': torch.from_numpy(_to_unit(d['ctrlpts'], self.gripper_box)).float(),
                'scores': torch.from_numpy(scores).float(),
                'input_ori': torch.from_numpy(np.asarray(d['obj_theta']) / np.pi - 1.0).float(),
                'input_pos': torch.from_numpy(np.asarray(d['obj_pos'])[..., :2] / 0.03).float(),
                'object_vertices': verts}
