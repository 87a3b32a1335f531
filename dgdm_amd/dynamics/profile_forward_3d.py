"""3-D dynamics model with the reference's interface (dynamics/profile_forward_3d.py:13-86) on the HIP path."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import engine
from ._backed import HipBacked
from .models.pointnet2 import PointNet2, draw_fps_starts
from .profile_forward_2d import _mlp2, _trunk, get_embedder, timestep_embedding  # noqa: F401  (re-exported like the reference)


class ProfileForward3DModel(HipBacked):
    def __init__(self, W=256, params_ch=1250, ori_ch=1, pos_ch=2, output_ch=3):
        super().__init__()
        if W != 256 or output_ch != 3 or ori_ch != 1 or pos_ch != 2:
            raise NotImplementedError("the HIP trunk is built for W=256, output_ch=3, ori_ch=1, pos_ch=2 (generator/train.py:86)")
        self.W, self.output_ch, self.params_ch = W, output_ch, params_ch
        self.ori_embed, self.ori_ch = get_embedder(ori_ch, 4, 0, scalar_factor=1)
        self.pos_embed, self.pos_ch = get_embedder(pos_ch, 4, 0, scalar_factor=1)
        self.pose_embed_dim = self.ori_ch + self.pos_ch
        self.time_embed_dim = self.object_encode_dim = self.gripper_encode_dim = W
        self.time_encoder = _mlp2(W // 2, W, nn.SiLU())        # present in checkpoints, never called (:83)
        self.object_encoder = PointNet2(W)
        self.gripper_encoder = _mlp2(params_ch, W, nn.ReLU())
        self.linears = _trunk(3 * W + self.pose_embed_dim, [2 * W] + [W] * 7)
        self.output = nn.Linear(W, output_ch)

    def _build_handle(self):
        return engine.Dynamics(3, self.plain_state_dict(), self.params_ch)

    def forward(self, x_ctrl, x_ori, x_pos, timesteps=None, object_vertices=None):
        """ctrlpts [rows,3,params_ch] (channel 1 is used), ori [rows,1], pos [rows,2], timesteps [rows],
        object points [rows,3,N] -> [rows,3].  Draws the FPS starts from the torch CPU generator like the reference."""
        s1, s2 = draw_fps_starts(object_vertices.shape[2], object_vertices.shape[0])
        return self.handle().forward3d(x_ctrl, x_ori, x_pos, timesteps, object_vertices, s1, s2)
