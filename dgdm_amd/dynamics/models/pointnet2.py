"""``PointNet2`` SSG object encoder with the reference's interface (dynamics/models/pointnet2.py:11-32)."""
from __future__ import annotations

import torch

from ... import engine, synth
from .._backed import HipBacked
from .pointnet2_utils import PointNetSetAbstraction


def draw_fps_starts(n_points: int, rows: int):
    """The two ``torch.randint`` draws one forward consumes from the CPU generator: sa1 over N points, then sa2 over
    the 512 sa1 centres (dynamics/models/pointnet2_utils.py:83, called from :132 for each non-group-all layer)."""
    return torch.randint(0, n_points, (rows,), dtype=torch.long), torch.randint(0, 512, (rows,), dtype=torch.long)


class PointNet2(HipBacked):
    def __init__(self, num_output_ch, normal_channel=False):
        super().__init__()
        if normal_channel or num_output_ch != 256:
            raise NotImplementedError("the HIP encoder is built for xyz-only clouds and 256 output channels (profile_forward_3d.py:31)")
        self.normal_channel, self.num_output_ch = normal_channel, num_output_ch
        self.sa1 = PointNetSetAbstraction(512, 0.2, 32, 3, [64, 128], False)
        self.sa2 = PointNetSetAbstraction(128, 0.4, 64, 128 + 3, [128, num_output_ch], False)
        self.sa3 = PointNetSetAbstraction(None, None, None, 256 + 3, [num_output_ch], True)

    def _build_handle(self):
        # a stand-alone encoder borrows a 3-D model handle whose other branches are zero
        full = {k: torch.zeros(shape) if shape else torch.zeros((), dtype=torch.int64) for k, shape in synth.dyn3d_spec(42)}
        full.update({"object_encoder." + k: v for k, v in self.plain_state_dict().items()})
        return engine.Dynamics(3, full, 42)

    def embed(self, handle: "engine.Dynamics", xyz: torch.Tensor) -> torch.Tensor:
        s1, s2 = draw_fps_starts(xyz.shape[2], xyz.shape[0])
        return handle.pointnet2(xyz, s1, s2)

    def forward(self, xyz):
        """xyz [B,3,N] -> (x [B,256], l3_points [B,256,1])."""
        x = self.embed(self.handle(), xyz)
        return x, x.reshape(x.shape[0], self.num_output_ch, 1)
