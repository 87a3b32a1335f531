"""Trainer.step / Trainer.inference of the 2-D dynamics model on the GPU (csrc/train2d.hip through dgdm_amd.dynamics.trainer)
against the reference's own Trainer (tests/golden/g10_train2d.npz) and against the oracle at other sizes."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import dgdm_oracle as orc
from tests import util
from tests.test_oracle_golden import BN_FED_BIAS, _train_case, check_training

pytestmark = pytest.mark.gpu


def _args(wd, L=14, nv=100, T=15, lr=1e-4):
    return argparse.Namespace(use_sub_batch=False, sub_bs=1024, grid_size=360, learning_rate=lr, weight_decay=wd, num_epochs=100,
                              checkpoint_path=None, fingers_3d=False, ctrlpts_dim=L, object_max_num_vertices=nv,
                              num_timesteps_per_batch=1, num_inference_steps=5, num_train_timesteps=T)


def dp_data(seed, rows, L=14, nv=100):
    sd = util.synth.synth_state_dict(util.synth.dyn2d_spec(L, 2 * nv), 50 + seed)
    rs = np.random.RandomState(seed)
    data = [torch.from_numpy(rs.uniform(-1, 1, s).astype(np.float32)) for s in ((rows, L), (rows, 3), (rows, 1), (rows, 2), (rows, 2 * nv))]
    return sd, data


class _HipTrainer:
    def __init__(self, sd, T, wd):
        from dynamics.trainer import Trainer
        self.t = Trainer(_args(wd, T=T))
        self.t.create_model(state_dict=sd)

    def lr_step(self):
        self.t.lr_scheduler.step()

    def step(self, *a):
        return self.t.step(*a)

    def inference(self, *a):
        return self.t.inference(*a)

    def gradients(self):
        return self.t.gradients()

    def state_dict(self):
        return self.t.state_dict()


@pytest.fixture(scope="module")
def dev():
    from dgdm_amd import _lib
    _lib.device_init(0)
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["wd0", "wd1"])
def test_trainer2d_matches_reference(dev, tag):
    """Three training steps (cosine schedule stepped once) and one eval call: losses, predictions, the gradients of step 1, every
    parameter after step 3 and the BatchNorm running statistics against the reference's own Trainer on the same draws."""
    g = util.load("g10_train2d.npz")
    rec = _train_case(g, tag, _HipTrainer)
    worst = check_training(g, tag, rec, 2e-5, 5e-5, 4e-6)
    print(tag, "largest gradient error / tensor rms:", max(worst.values()))
    # Trainer.inference (eval mode: running statistics) on the weights this run ended with, against the oracle on the same weights
    n_g, n_p, L, nv, T = [int(v) for v in g["dims"]]
    data = util.train2d_data(int(g["data_seed"]), n_g, n_p)
    o = orc.Trainer2D(rec["final"], T, 1e-4)
    torch.manual_seed(9)
    po, lo = o.inference(*data)
    torch.manual_seed(9)
    ph, lh = rec["trainer"].inference(*data)
    assert util.rel_l2(ph.cpu(), po) < 2e-5 and abs(lh / lo - 1) < 2e-5


def _one_step(rows, L, nv, seed):
    sd = util.synth.synth_state_dict(util.synth.dyn2d_spec(L, 2 * nv), 50 + seed)
    rs = np.random.RandomState(seed)
    data = [torch.from_numpy(rs.uniform(-1, 1, s).astype(np.float32)) for s in ((rows, L), (rows, 3), (rows, 1), (rows, 2), (rows, 2 * nv))]
    o = orc.Trainer2D(sd, 15, 1e-4)
    torch.manual_seed(77)
    lo, po = o.step(*data)
    return sd, data, o, lo, po


def _hip_step(sd, data, L, nv):
    from dynamics.trainer import Trainer
    t = Trainer(_args(0.0, L, nv))
    t.create_model(state_dict=sd)
    torch.manual_seed(77)
    lh, ph = t.step(*data)
    return lh, ph.cpu(), t.gradients(), t.state_dict()


def test_trainer2d_ragged_rows_vs_oracle(dev):
    """Row counts that are no multiple of any tile (130, 200), another control-point and object size (L = 42, 64 vertices), one step
    each: loss, predictions, every gradient and the running statistics against the oracle on the same draws; the same step twice
    gives the same bits.  The batch is the first of its seeds whose ReLU inputs all stay 5e-6 from zero in the oracle's run (see
    make_golden.g10_train2d: a ReLU input that rounds to the other side in another summation order changes a whole BatchNorm
    column of the backward pass, in the reference as much as here)."""
    for rows, L, nv in ((130, 42, 64), (200, 14, 100)):
        for seed in range(200):
            sd, data, o, lo, po = _one_step(rows, L, nv, seed)
            if o.relu_margin >= 5e-6:
                break
        assert o.relu_margin >= 5e-6
        outs = [_hip_step(sd, data, L, nv) for _ in range(2)]
        lh, ph, gh, sh = outs[0]
        assert abs(lh / lo - 1) < 1e-5 and util.rel_l2(ph, po) < 2e-5
        for k, ref in o.grads.items():
            if k in BN_FED_BIAS:
                continue
            err = float((gh[k].double() - ref.double()).abs().max() / ref.double().pow(2).mean().sqrt())
            # (the time encoder sees 15 distinct inputs: its gradients are sums that cancel ten-fold, and carry that much more rounding)
            assert err < (2e-4 if k.startswith('time_encoder') else 6e-5), (rows, k, err)
        for k in sh:
            assert torch.equal(sh[k], outs[1][3][k]), k
        assert torch.equal(ph, outs[1][1]) and lh == outs[1][0]
        for k in ("linears.1.running_mean", "linears.22.running_var"):
            assert float((sh[k] - o.sd[k]).abs().max()) < 1e-5, k


def test_trainer2d_many_rows_vs_oracle(dev):
    """4099 rows (33 row tiles, 17 weight-gradient splits): forward exact to float32 rounding; the gradients within what a handful of
    ReLU inputs at rounding distance from zero can move them (1e-2 of the tensor norm: a million ReLU inputs per layer, about one of
    them within 1e-6 of zero; an indexing error at a tile or split edge would show as O(0.1)); the output layer's - the only
    gradients above every ReLU mask - tight."""
    sd, data, o, lo, po = _one_step(4099, 14, 100, 1)
    lh, ph, gh, sh = _hip_step(sd, data, 14, 100)
    assert abs(lh / lo - 1) < 1e-5 and util.rel_l2(ph, po) < 2e-5
    for k in ("output.weight", "output.bias"):
        assert util.rel_l2(gh[k], o.grads[k]) < 2e-5, k
    for k, ref in o.grads.items():
        if k not in BN_FED_BIAS:
            assert util.rel_l2(gh[k], ref) < 1e-2, (k, util.rel_l2(gh[k], ref))


def test_trainer2d_checkpoint_round_trip(dev, tmp_path):
    """save_checkpoint writes the DataParallel key layout the sampling path loads (trainer.py:105-106 -> generator/train.py:90), and the
    eval-mode forward of the saved weights equals Trainer.inference's predictions."""
    from dynamics.trainer import Trainer
    from dynamics.profile_forward_2d import ProfileForward2DModel
    sd = util.dyn2d_sd(41, 100)
    data = util.train2d_data(5)
    t = Trainer(_args(0.0))
    t.create_model(state_dict=sd)
    torch.manual_seed(1)
    for _ in range(2):
        t.step(*data)
    path = str(tmp_path / "dyn.pt")
    t.save_checkpoint(path)
    ck = torch.load(path)
    assert all(k.startswith("module.") for k in ck) and int(ck["module.linears.1.num_batches_tracked"]) == 2
    torch.manual_seed(2)
    pred, loss = t.inference(*data)
    torch.manual_seed(2)
    noise = torch.randn((384, 14))
    ts = torch.randint(0, 15, (384,)).long()
    noisy = orc.DDIM(15).add_noise(data[0], noise, ts)
    m = ProfileForward2DModel(params_ch=14, object_ch=200)
    m.load_state_dict({k[len("module."):]: v for k, v in ck.items()})
    out = m.to(dev)(noisy.to(dev), data[2].to(dev), data[3].to(dev), (ts.float() / 15).to(dev), data[4].to(dev))
    assert util.rel_l2(out.cpu(), pred.cpu()) < 2e-5
    # a single row: fine in eval mode, refused in training mode as nn.BatchNorm1d refuses it
    one = [d[:1] for d in data]
    p1, _ = t.inference(*one)
    assert p1.shape == (1, 3) and bool(torch.isfinite(p1).all())
    with pytest.raises(Exception):
        t.step(*one)


def test_training_driver_end_to_end(dev, tmp_path):
    """`python dynamics/main.py <flags of train_dynamics_2d.sh>` on a synthetic dataset in the simulator's file format (scores a smooth
    function of the inputs so there is something to learn): the validation loss falls, the log and the checkpoints appear, and
    best.pt - DataParallel key layout - loads into the sampling path's model (generator/train.py:88-90)."""
    import json
    import subprocess
    import sys
    from dynamics.profile_forward_2d import ProfileForward2DModel
    rs = np.random.RandomState(0)
    for split, n in (("train", 12), ("val", 4)):
        root = tmp_path / split
        root.mkdir()
        for i in range(n):
            cells = 48
            cy = rs.uniform(-0.045, 0.015, 14)
            th, pos = rs.uniform(0, 2 * np.pi, cells), rs.uniform(-0.03, 0.03, (cells, 3))
            d = {"ctrlpts": np.stack([np.linspace(-0.12, 0.12, 14), cy], 1), "obj_theta": th, "obj_pos": pos,
                 "delta_theta": 0.05 * np.sin(th) * (1 + 10 * cy.mean()), "delta_pos": 0.1 * pos[:, :2] + 0.002 * np.cos(th)[:, None],
                 "object_vertices": rs.uniform(-0.05, 0.05, (6, 2))}
            np.savez(root / f"s{i}.npz", d)
    save = tmp_path / "out"
    cmd = [sys.executable, "dynamics/main.py", f"--save_dir={save}", "--ctrlpts_dim=14", "--batch_size=4", "--object_max_num_vertices=100",
           f"--data_dir={tmp_path / 'train'}", f"--test_data_dir={tmp_path / 'val'}", "--learning_rate=1e-3", "--weight_decay=0", "--num_epochs=12",
           "--val_step=1", "--save_ckpt_step=1000", "--patience=100", "--num_workers=0", "--num_train_timesteps=15", "--num_inference_steps=5",
           "--num_timesteps_per_batch=1"]
    r = subprocess.run(cmd, cwd=str(util.GOLDEN + "/../.."), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    logs = [json.loads(l) for l in open(save / "log.jsonl")]
    val = [l["val/average loss"] for l in logs if "val/average loss" in l]
    assert len(val) == 12 and min(val[-3:]) < 0.7 * val[0], val
    assert (save / "best.pt").exists() and (save / "0_0.pt").exists()
    ck = torch.load(save / "best.pt")
    m = ProfileForward2DModel(params_ch=14, object_ch=200)
    missing = m.load_state_dict({k[len("module."):]: v for k, v in ck.items()})
    assert not missing.missing_keys and not missing.unexpected_keys
    out = m.to(dev)(torch.zeros(4, 14, device=dev), torch.zeros(4, 1, device=dev), torch.zeros(4, 2, device=dev), torch.zeros(4, device=dev),
                    torch.zeros(4, 200, device=dev))
    assert out.shape == (4, 3) and bool(torch.isfinite(out).all())


def test_trainer2d_data_parallel(dev, tmp_path):
    """Two ranks (sharing this box's GPU; gloo carries the gradient all-reduce) train data-parallel with nn.DataParallel's semantics
    (trainer.py:41-43): chunks as torch.chunk cuts them (51 + 50 rows), BatchNorm statistics per chunk, gradients summed, the same Adam
    step on every rank, running statistics of rank 0.  Against the oracle's statement of those semantics on the same draws, one step
    tight (a batch whose ReLU inputs stay clear of zero) and a second step + eval call loosely; and the draws do not depend on the
    rank count (DGDM_TORCH_SEED pins the CPU generator the reference leaves unseeded)."""
    import subprocess
    import sys
    rows = 101
    for seed in range(300):
        sd, data = dp_data(seed, rows)
        o = orc.Trainer2D(sd, 15, 1e-4)
        torch.manual_seed(4242)
        lo, po = o.step(*data, replicas=2)
        if o.relu_margin >= 5e-6:
            break
    assert o.relu_margin >= 5e-6
    g1 = {k: v.clone() for k, v in o.grads.items()}
    lo2, po2 = o.step(*data, replicas=2)
    pio, lio = o.inference(*data)          # eval mode with replica 0's running statistics, the stream of draws going on
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DGDM_TORCH_SEED="4242", DGDM_DIST_BACKEND="gloo")
    out = str(tmp_path / "dp.npz")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(root, "tests", "dp_train_worker.py"), str(seed), str(rows), "2", out],
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    g = np.load(out)
    assert int(g["world"]) == 2
    assert abs(float(g["loss0"]) / lo - 1) < 1e-5 and util.rel_l2(g["pred0"], po) < 2e-5
    # the second step's predictions depend on the first update (every parameter moved by about lr): agreement = the update was the same
    assert abs(float(g["loss1"]) / lo2 - 1) < 1e-4 and util.rel_l2(g["pred1"], po2) < 1e-3
    for k, ref in o.grads.items():        # gradients of the LAST step are what the trainer holds; compare loosely (second step), then
        if k not in BN_FED_BIAS:          # the running statistics, which only rank 0's chunk feeds
            assert util.rel_l2(g["grad/" + k], ref) < 2e-2, k
    for k in ("linears.1.running_var", "linears.22.running_var"):
        assert float(np.abs(g["sd/" + k] - o.sd[k].numpy()).max()) < 1e-4, k
    assert int(g["sd/linears.1.num_batches_tracked"]) == 2
    # the eval call after the two steps: EVERY rank's chunk is evaluated with rank 0's running statistics (nn.DataParallel keeps replica
    # 0's buffers; they are broadcast after each step) - with per-rank statistics the second chunk's predictions are off by per cents
    assert util.rel_l2(g["inf_pred"], pio) < 2e-3 and abs(float(g["inf_loss"]) / lio - 1) < 2e-3, (util.rel_l2(g["inf_pred"], pio), float(g["inf_loss"]), lio)


def test_trainer2d_grouped_encoders(dev):
    """The time encoder on its 15 distinct inputs and the object encoder once per sample (dgdm_trainer2d_set_groups) give the step
    every-row evaluation gives: same loss and predictions to float32 rounding, gradients within the summation-order noise (the two
    encoders' gradients are summed per group first), and the object hint is ignored when the rows are no multiple of the run."""
    from dynamics.trainer import Trainer
    n_g, n_p = 5, 40
    sd = util.dyn2d_sd(41, 100)
    data = util.train2d_data(9, n_g, n_p)
    res = []
    for grouped, rps in ((False, None), (True, None), (True, n_p), (True, 7)):
        t = Trainer(_args(0.0))
        t.create_model(state_dict=sd)
        t.group_encoders = grouped
        torch.manual_seed(5)
        loss, pred = t.step(*data, rows_per_sample=rps)
        torch.manual_seed(6)
        pi, li = t.inference(*data, rows_per_sample=rps)
        res.append((loss, pred.cpu(), t.gradients(), pi.cpu(), li))
    base = res[0]
    for loss, pred, grads, pi, li in res[1:]:
        assert abs(loss / base[0] - 1) < 1e-6 and util.rel_l2(pred, base[1]) < 1e-6
        assert abs(li / base[4] - 1) < 1e-4 and util.rel_l2(pi, base[3]) < 1e-4          # after one Adam step on gradients that differ by rounding
        for k, ref in base[2].items():
            if k not in BN_FED_BIAS:
                assert util.rel_l2(grads[k], ref) < (2e-4 if "encoder" in k else 2e-5), (k, util.rel_l2(grads[k], ref))
    assert all(torch.equal(res[1][2][k], res[3][2][k]) for k in res[1][2])          # rows_per_sample = 7 does not divide 200 rows: ignored


def test_trainer2d_sub_batches(dev):
    """--use_sub_batch (trainer.py:81-94): the draws once for the whole batch, one optimizer step per slice; equals stepping a plain
    trainer slice by slice with the same draws, and returns the reference's loss average."""
    from dynamics.trainer import Trainer
    sd = util.dyn2d_sd(41, 100)
    data = util.train2d_data(9, 3, 40)           # 120 rows, slices of 50, 50, 20
    a = _args(0.0)
    a.use_sub_batch, a.sub_bs = True, 50
    t = Trainer(a)
    t.create_model(state_dict=sd)
    torch.manual_seed(3)
    loss, pred = t.step(*data)
    ref = Trainer(_args(0.0))
    ref.create_model(state_dict=sd)
    torch.manual_seed(3)
    noise, ts = ref._draw(120)
    losses, preds = [], []
    for i in range(0, 120, 50):
        l, p = ref._run(*[d[i:i + 50] for d in (data[0], data[1], data[2], data[3], data[4])], True, None, (noise[i:i + 50], ts[i:i + 50]))
        losses.append(l)
        preds.append(p)
    assert torch.equal(pred, torch.cat(preds)) and abs(loss - sum(losses) / (120 / 50)) < 1e-7
    assert all(torch.equal(v, ref.state_dict()[k]) for k, v in t.state_dict().items())
