"""Host side of the dynamics-model training driver (dgdm_amd/dynamics/{dataloader,main}.py): CPU only."""
import numpy as np
import torch

from tests import util


def test_dataset_matches_reference(tmp_path):
    """DynamicsDataset on the synthetic files of the fixture against what the reference's own DynamicsDataset returned for them."""
    from dynamics.dataloader import DynamicsDataset
    g = util.load("g11_dataset.npz")
    util.write_synth_dataset(str(tmp_path), int(g["seed"]))
    ds = DynamicsDataset(str(tmp_path), object_max_num_vertices=8)
    assert len(ds) == int(g["n"]) and np.allclose(ds.threshold / ds.std, g["threshold_std"], rtol=0, atol=0)
    for i in range(len(ds)):
        item = ds[i]
        assert set(item) == {"ctrlpts", "scores", "input_ori", "input_pos", "object_vertices"}
        for k, v in item.items():
            assert v.dtype == torch.float32 and np.array_equal(v.numpy(), g[f"{i}/{k}"]), (i, k)


def test_batch_rows_and_accuracy(tmp_path):
    """A batch [samples, cells, ...] becomes rows (sample, cell): the sample's control ordinates and flattened vertices on each of
    its rows (dynamics/main.py:25-35); the three-way class accuracy of dynamics/main.py:37-39."""
    from dgdm_amd.dynamics.main import batch_rows, class_accuracy
    from dynamics.dataloader import DynamicsDataset
    util.write_synth_dataset(str(tmp_path), 3, n_files=4, cells=6)
    ds = DynamicsDataset(str(tmp_path), object_max_num_vertices=8)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)))
    ctrl, score, ori, pos, obj = batch_rows(batch)
    assert ctrl.shape == (24, 14) and score.shape == (24, 3) and ori.shape == (24, 1) and pos.shape == (24, 2) and obj.shape == (24, 16)
    for s in range(4):
        for c in range(6):
            r = s * 6 + c
            assert torch.equal(ctrl[r], batch["ctrlpts"][s, :, 1]) and torch.equal(obj[r], batch["object_vertices"][s].reshape(-1))
            assert torch.equal(score[r], batch["scores"][s, c]) and float(ori[r]) == float(batch["input_ori"][s, c])
            assert torch.equal(pos[r], batch["input_pos"][s, c])
    t = [0.5, 0.5, 0.5]
    sc = torch.tensor([[1.0, 0.0, -1.0], [0.0, 0.0, 0.0], [-1.0, 1.0, 1.0], [0.6, -0.6, 0.4]])
    pr = torch.tensor([[0.7, 0.1, -0.2], [0.6, 0.0, 0.0], [-2.0, 0.4, 1.0], [0.4, -0.7, 0.6]])
    ref = [float(np.mean([(2 if a > t[j] else 0 if a < -t[j] else 1) == (2 if b > t[j] else 0 if b < -t[j] else 1) for a, b in zip(sc[:, j], pr[:, j])]))
           for j in range(3)]
    assert class_accuracy(sc, pr, t) == ref
