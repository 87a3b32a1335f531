import sys; sys.path.insert(0, '.')
import torch, numpy as np
from oracle import dgdm_oracle as orc
from tests import util
print(torch.__config__.show().split("\n")[2:6], torch.get_num_threads())
import subprocess; print(subprocess.run("lscpu | grep 'Model name'", shell=True, capture_output=True, text=True).stdout)
ctrl, score, ori, pos, obj = util.train3d_data(7)
xyz = obj.permute(0, 2, 1).contiguous()
R = xyz.shape[0]
torch.manual_seed(5); s1 = torch.randint(0, 512, (R,))
nx = orc._gather(xyz, orc.farthest_point_sample(xyz, 512, s1))
mm = torch.matmul(nx, xyz.permute(0, 2, 1)).numpy()
a = nx.numpy().astype(np.float64)[:, :, None, :]; b = xyz.numpy().astype(np.float64)[:, None, :, :]
f32 = lambda x: x.astype(np.float32).astype(np.float64)
fma = lambda x, y, c: f32(x * y + c)
cands = {"fma chain 0,1,2": fma(a[..., 2], b[..., 2], fma(a[..., 1], b[..., 1], f32(a[..., 0] * b[..., 0]))),
         "unfused ((0+1)+2)": f32(f32(f32(a[..., 0] * b[..., 0]) + f32(a[..., 1] * b[..., 1])) + f32(a[..., 2] * b[..., 2])),
         "fma chain 2,1,0": fma(a[..., 0], b[..., 0], fma(a[..., 1], b[..., 1], f32(a[..., 2] * b[..., 2]))),
         "exact then round": f32(a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2])}
for k, v in cands.items():
    print(k, "mismatching elements:", int((v.astype(np.float32) != mm).sum()), "of", mm.size)
for nt in (1, 4, 32):
    torch.set_num_threads(nt)
    m2 = torch.matmul(nx, xyz.permute(0, 2, 1)).numpy()
    print("threads", nt, "vs default:", int((m2 != mm).sum()))
m1 = torch.cat([torch.matmul(nx[i:i+1], xyz[i:i+1].permute(0, 2, 1)) for i in range(R)]).numpy()
print("batch 1 vs batch 8:", int((m1 != mm).sum()))
