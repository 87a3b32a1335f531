"""``DynamicsDataset`` of the dynamics-model training driver (reference: dynamics/dataloader.py:7-78): one ``.npz`` per simulated
(gripper, object) pair holding the pose grid it was rolled out on.  Host code.  2-D: the object's contour vertices ride in the file.
3-D (--fingers_3d, :57-67): the file names its object; the reference samples ``object_max_num_vertices`` points from
``<object_mesh_dir>/<name>/model.obj`` with open3d (dynamics/utils.py, asset tooling outside this package: open3d is not in the image) -
here the points are read from ``<object_mesh_dir>/<name>/points.npy`` ([n >= object_max_num_vertices, 3] metres, the first
object_max_num_vertices rows are used) or from an ``object_points`` entry of the data file itself.

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
# dynamics/main.py:64-71,83-85 (--fingers_3d)
GRIPPER_BOX_3D = ((-0.12, 0.12), (-0.1, 0.0), (0.0, 0.12))
OBJECT_BOX_3D = ((-0.1, 0.1), (-0.1, 0.1), (0.0, 0.12))


def _to_unit(a: np.ndarray, box: Sequence[Sequence[float]]) -> np.ndarray:
    out = np.array(a, dtype=np.float64, copy=True)
    for axis, (lo, hi) in enumerate(box):
        out[..., axis] = (out[..., axis] - lo) / (hi - lo) * 2.0 - 1.0
    return out


class DynamicsDataset(Dataset):
    def __init__(self, dataset_dir: str, object_max_num_vertices: int = 10, fingers_3d: bool = False, gripper_box=None,
                 object_box=None, object_mesh_dir: str = "", **unused):
        self.fingers_3d = fingers_3d
        self.std, self.threshold = SCORE_STD[0 if fingers_3d else 1], SCORE_THRESHOLD[0 if fingers_3d else 1]
        self.gripper_box = gripper_box or (GRIPPER_BOX_3D if fingers_3d else GRIPPER_BOX_2D)
        self.object_box = object_box or (OBJECT_BOX_3D if fingers_3d else OBJECT_BOX_2D)
        self.object_max_num_vertices, self.object_mesh_dir, self.object_pts = object_max_num_vertices, object_mesh_dir, {}
        self.data_files = sorted(os.path.join(root, f) for root, _, files in os.walk(dataset_dir) for f in files if f.endswith('.npz'))

    def __len__(self) -> int:
        return len(self.data_files)

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        d = np.load(self.data_files[idx], allow_pickle=True)['arr_0'].item()
        scores = np.stack([d['delta_theta'] / self.std[0], d['delta_pos'][:, 0] / self.std[1], d['delta_pos'][:, 1] / self.std[2]], axis=1)
        if self.fingers_3d:
            name = str(d['object_name']) if 'object_name' in d else "object"
            if name not in self.object_pts:                                               # cached per object name (:58-66)
                f = os.path.join(self.object_mesh_dir or "", name, 'points.npy')
                if os.path.isfile(f):
                    pts = np.load(f)
                elif 'object_points' in d:
                    pts = np.asarray(d['object_points'])
                else:
                    raise FileNotFoundError(f"3-D object '{name}': neither {f} nor an 'object_points' entry in {self.data_files[idx]} (the reference samples "
                                            "the points from model.obj with open3d, which this package does not ship)")
                self.object_pts[name] = _to_unit(np.asarray(pts)[:self.object_max_num_vertices, :3], self.object_box)
            verts = torch.from_numpy(self.object_pts[name]).float()
        else:
            verts = torch.from_numpy(_to_unit(d['object_vertices'], self.object_box)).float()
            verts = torch.cat([verts, torch.zeros(self.object_max_num_vertices - verts.shape[0], 2)], dim=0)       # zero-padded (:73)
        return {'ctrlpts': torch.from_numpy(_to_unit(d['ctrlpts'], self.gripper_box)).float(),
                'scores': torch.from_numpy(scores).float(),
                'input_ori': torch.from_numpy(np.asarray(d['obj_theta']) / np.pi - 1.0).float(),
                'input_pos': torch.from_numpy(np.asarray(d['obj_pos'])[..., :2] / 0.03).float(),
                'object_vertices': verts}
