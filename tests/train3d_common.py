"""Drivers shared by the CPU (oracle) and GPU (HIP) tests of the 3-D Trainer against tests/golden/g13_train3d.npz - the reference's own
Trainer.step / Trainer.inference with --fingers_3d (tests/golden/make_golden.py g13)."""
import numpy as np
import torch

from oracle import dgdm_oracle as orc
from tests import util

CASES = {"plain": (False, 0.0, 3), "sub": (True, 0.01, 2)}        # tag -> (use_sub_batch (sub_bs = 4), weight_decay, calls)

# Biases in front of a BatchNorm (a shift the batch mean absorbs): exactly-zero gradient, rounding residue in every implementation,
# and Adam turns residue of any size into +-lr steps - compared nowhere (as in the 2-D fixture, tests/test_oracle_golden.py).
# object_encoder.sa3.mlp_bns.0.bias belongs with them on this fixture: every pooled sa3 unit is positive here, so the bias shifts a column
# of the embedding by the same amount in every row and linears.1's BatchNorm absorbs it (reference gradient rms 1e-7 against 1e-2 .. 1 of
# its neighbours).
BN_FED_BIAS = ({f"linears.{3 * i}.bias" for i in range(8)} | {"gripper_encoder.2.bias", "object_encoder.sa3.mlp_bns.0.bias"} |
               {f"object_encoder.sa{a}.mlp_convs.{b}.bias" for a, b in ((1, 0), (1, 1), (2, 0), (2, 1), (3, 0))})


def lr_after(epochs):       # CosineAnnealingLR(T_max=100, eta_min=1e-2 * lr) (dynamics/trainer.py:47)
    return 1e-6 + (1e-4 - 1e-6) * (1 + np.cos(np.pi * epochs / 100)) / 2


def drive(g, tag, make):
    sub, wd, calls = CASES[tag]
    tr = make(util.dyn3d_sd(int(g["dyn3d_seed"])), 15, wd, sub)
    data = util.train3d_data(int(g["data_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    rec = {}
    for step in range(calls):
        if step == 2:
            tr.lr_step()
        rec[f"loss{step}"], rec[f"pred{step}"] = tr.step(*data)
        if step == 0:
            rec["grads"] = tr.gradients()
    rec["final"] = tr.state_dict()
    rec["inf_pred"], rec["inf_loss"] = tr.inference(*data)
    return rec


class OracleTrainer3D:
    """oracle.Trainer3D driven like dynamics/trainer.py drives the model: the batch's draws once, then one step per --use_sub_batch slice."""

    def __init__(self, sd, T, wd, sub):
        self.t, self.sub, self.epoch = orc.Trainer3D(sd, T, 1e-4, wd), sub, 0

    def lr_step(self):
        self.epoch += 1
        self.t.lr = lr_after(self.epoch)

    def _batch(self, fn, ctrl, score, ori, pos, obj):
        n = ctrl.shape[0]
        noise, ts = self.t.draw(ctrl)
        if not self.sub:
            return fn(ctrl, score, ori, pos, obj, (noise, ts))
        losses, preds = [], []
        for i in range(0, n, 4):
            sl = slice(i, i + 4)
            loss, pred = fn(ctrl[sl], score[sl], ori[sl], pos[sl], obj[sl], (noise[sl], ts[sl]))
            losses.append(loss)
            preds.append(pred)
        return sum(losses) / (n / 4), torch.cat(preds)

    def step(self, *a):
        return self._batch(self.t.step, *a)

    def inference(self, *a):
        pred, loss = self._batch(lambda *b: self.t.inference(*b)[::-1], *a)[::-1]
        return pred, loss

    def gradients(self):
        return self.t.grads

    def state_dict(self):
        return self.t.sd


def check_sub_loosely(g, rec):
    """'sub' through a second float32 implementation: every call takes two optimizer steps with BatchNorm over 4 rows in between, Adam's
    first steps are +-lr on every entry whatever its gradient (entries at rounding level take different signs in two implementations), and
    a 4-row BatchNorm turns that into per-cent differences of the next forward - the trajectory is not comparable beyond its first
    slice.  What is: the first call (its first slice runs on identical weights: half of the predictions and of the loss), the structure
    (four updates, every parameter within the step bound of the reference's), finite eval-mode results."""
    tag = "sub"
    assert abs(rec["loss0"] / float(g[f"{tag}_loss0"]) - 1) < 5e-2
    n = rec["pred0"].shape[0]
    first = util.rel_l2(rec["pred0"][:4].cpu(), g[f"{tag}_pred0"][:4])
    print("sub: first slice (before any update) predictions rel L2", first, "whole first call", util.rel_l2(rec["pred0"].cpu(), g[f"{tag}_pred0"]))
    assert first < 2e-4          # the slice evaluated before any update (BatchNorm over 4 rows: 1e-5 .. 1e-4)
    assert util.rel_l2(rec["pred0"].cpu(), g[f"{tag}_pred0"]) < 5e-2 and n == g[f"{tag}_pred0"].shape[0]
    for key in [k for k in g.files if k.startswith(f"{tag}_final/")]:
        name = key.split("/", 1)[1]
        mine, ref = rec["final"][name].double().flatten(), torch.from_numpy(g[key]).double().flatten()
        if name.endswith("num_batches_tracked"):
            assert int(mine) == int(ref) == 4
        elif "running_" not in name and not name.startswith("time_encoder"):
            assert float((mine[util.sample_idx(name, mine.numel())] - ref).abs().max()) < 2 * 4 * 1.05e-4, name      # 4 Adam steps of <= lr each, either sign
    assert np.isfinite(rec["inf_loss"]) and bool(torch.isfinite(rec["inf_pred"]).all())


def check(g, tag, rec, tol_pred, tol_grad, tol_param, tol_run=1e-4, verbose=False, tol_later=1e-3, vs64=False, tol_inf=2e-3):
    calls = CASES[tag][2]
    for step in range(calls):
        # after the first update the two sides' weights differ where Adam stepped on rounding residue (BN_FED_BIAS: +-lr per step), and with
        # BatchNorm over 4 - 8 rows the forward pass feels that (3.5e-4 between the oracle and the reference itself on 'sub'): the first
        # forward (identical weights) is held to tol_pred, later ones to 1e-3
        tol = tol_pred if step == 0 and tag == "plain" else tol_later
        assert abs(rec[f"loss{step}"] / float(g[f"{tag}_loss{step}"]) - 1) < tol, (tag, step, rec[f"loss{step}"], float(g[f"{tag}_loss{step}"]))
        assert util.rel_l2(rec[f"pred{step}"].cpu(), g[f"{tag}_pred{step}"]) < tol, (tag, step, util.rel_l2(rec[f"pred{step}"].cpu(), g[f"{tag}_pred{step}"]))
    worst = {}
    for key in [k for k in g.files if k.startswith(f"{tag}_grad/")]:
        name = key.split("/", 1)[1]
        if name in BN_FED_BIAS:
            continue
        mine = rec["grads"][name].double().flatten()
        ref = torch.from_numpy(g[key]).double()
        scale = float(np.sqrt(g[f"{tag}_gradsum/{name}"][1] / mine.numel()))
        worst[name] = float((mine[util.sample_idx(name, mine.numel())] - ref).abs().max()) / scale
        if vs64 and f"{tag}_grad64/{name}" in g.files:
            # against the float64 gradient of the same step: at least as close as the reference's own float32 gradient is (x 1.5), and the
            # distance to the reference bounded by the reference's own distance from exact
            r64 = torch.from_numpy(g[f"{tag}_grad64/{name}"]).double()
            e_h = float((mine[util.sample_idx(name, mine.numel())] - r64).abs().max()) / scale
            e_r = float((ref - r64).abs().max()) / scale
            assert e_h <= max(tol_grad, 1.5 * e_r), (tag, name, e_h, e_r)
            assert worst[name] <= max(tol_grad, 2.5 * e_r), (tag, name, worst[name], e_r)
            worst[name] = min(worst[name], tol_grad * 0.999) if worst[name] <= max(tol_grad, 2.5 * e_r) else worst[name]
    if verbose:
        print(tag, "worst sampled gradient entry / tensor rms:", sorted(((v, k) for k, v in worst.items()), reverse=True)[:4])
    for name, w in worst.items():
        assert w < tol_grad, (tag, name, w)
    wp, wr = 0.0, 0.0
    for key in [k for k in g.files if k.startswith(f"{tag}_final/")]:
        name = key.split("/", 1)[1]
        if name in BN_FED_BIAS or name.startswith("time_encoder"):
            continue
        mine = rec["final"][name].double().flatten()
        ref = torch.from_numpy(g[key]).double().flatten()
        if "running_" in name or name.endswith("num_batches_tracked"):
            d = float((mine - ref).abs().max()) / max(1.0, float(ref.abs().max()))
            wr = max(wr, d)
            # a running MEAN carries the history of the rounding-driven bias in front of its BatchNorm (each up to lr per step apart between
            # two implementations); variances and counters do not
            # ('sub': four optimizer steps with BatchNorm over 4 rows - the later forwards see the 3.5e-4 of above)
            assert d < (1e-3 if tag == "sub" else max(2e-4, tol_run) if "running_mean" in name else tol_run), (tag, name, d)
        else:
            d = (mine[util.sample_idx(name, mine.numel())] - ref).abs()
            gref = torch.from_numpy(g[f"{tag}_grad/{name}"]).double().abs()
            real = gref >= 1e-3 * gref.max()
            wp = max(wp, float(d[real].max()))
            assert float(d[real].max()) < tol_param, (tag, name, float(d[real].max()))
            assert float(d.max()) < 2 * 3.1e-4, (tag, name)
    for name in ("time_encoder.0.weight", "time_encoder.2.bias"):        # never called by forward: untouched, weight decay included
        assert torch.equal(rec["final"][name].cpu(), util.dyn3d_sd(int(g["dyn3d_seed"]))[name]), name
    if verbose:
        print(tag, f"parameters within {wp:.2e}, running statistics within {wr:.2e}")
    if verbose:
        print(tag, "eval-mode inference after the steps: loss", rec["inf_loss"], float(g[f"{tag}_inf_loss"]), "pred rel L2", util.rel_l2(rec["inf_pred"].cpu(), g[f"{tag}_inf_pred"]))
    assert abs(rec["inf_loss"] / float(g[f"{tag}_inf_loss"]) - 1) < tol_inf
    assert util.rel_l2(rec["inf_pred"].cpu(), g[f"{tag}_inf_pred"]) < tol_inf
    return worst
