"""The oracle against the golden vectors captured from the reference (tests/golden/make_golden.py).

This is what pins oracle/dgdm_oracle.py: same seeds, same FPS start indices -> same numbers as the
reference's own modules, to float32 round-off (torch CPU kernels are identical here, so most are exact).
"""
import numpy as np
import pytest
import torch

from oracle import dgdm_oracle as orc
from tests import util
from tests.golden.make_golden_names import OBJ16

TOL = 2e-6


def test_ddim_closed_forms():
    # SURVEY.md §8(a13,c): the only anchors that exist for the un-vendored scheduler (parity unpinned)
    s = orc.DDIM(15)
    s.set_timesteps(5)
    assert s.timesteps.tolist() == [12, 9, 6, 3, 0]
    assert abs(float(s.alphas_cumprod[0]) - 0.98668) < 1e-5
    assert abs(float(s.alphas_cumprod[14]) / 1.0756e-5 - 1) < 1e-3
    assert bool((s.alphas_cumprod[1:] < s.alphas_cumprod[:-1]).all())
    assert float(s.betas.max()) <= 0.999 + 1e-7
    x = torch.randn(3, 14, 1)
    e = torch.randn(3, 14, 1)
    last = s.step(e, 0, x)          # prev_t < 0 -> alpha_prev = 1 -> returns clamp(x0)
    a0 = s.alphas_cumprod[0]
    assert torch.allclose(last, ((x - (1 - a0) ** 0.5 * e) / a0 ** 0.5).clamp(-1, 1))
    s2 = orc.DDIM(1000)
    s2.set_timesteps(100)
    assert s2.timesteps[:3].tolist() == [990, 980, 970] and len(s2.timesteps) == 100
    n = s.add_noise(x, e, torch.tensor([5, 5, 5]))
    assert torch.allclose(n, s.alphas_cumprod[5] ** 0.5 * x + (1 - s.alphas_cumprod[5]) ** 0.5 * e)


def test_unet_matches_reference():
    g = util.load("g2_unet.npz")
    sd = util.unet_sd(g["seed"])
    for L in (14, 42):
        x = torch.from_numpy(g[f"x_L{L}"])
        for t in (0, 3, 12, 999):
            y = orc.unet1d_forward(sd, x, torch.full((4,), t, dtype=torch.int64))
            assert util.rel_l2(y, g[f"y_L{L}_t{t}"]) < TOL


def test_dyn2d_forward_and_grads():
    g = util.load("g3_dyn2d.npz")
    B, G, P, L, T, S, nv = g["dims"]
    sd = util.dyn2d_sd(g["seed"], nv)
    f = lambda k: torch.from_numpy(g[k])
    y = orc.dyn2d_forward(sd, f("fwd_xc"), f("fwd_xo"), f("fwd_xp"), f("fwd_t"), f("fwd_obj"))
    assert util.rel_l2(y, g["fwd_logits"]) < TOL
    s = util.setup('point', None, sd, T, S, L, G, P)
    x, obj = f("x"), f("obj")
    t = torch.full((int(B),), 9, dtype=torch.int64)
    for o in OBJ16:
        for name, rng in (("full", (-1.0, 1.0)), ("half", (-0.5, 0.25))):
            gr = orc.cond_fn(s, x, t, o, obj, rng, centers=f("centers") if o == 'convergence' else None)
            assert util.rel_l2(gr, g[f"grad_{o}_{name}"]) < 1e-5, (o, name)


def test_pointnet_indices_and_embedding():
    g = util.load("g4_pointnet.npz")
    sd = util.dyn3d_sd(g["seed"])
    clouds = torch.from_numpy(g["clouds"])
    st = torch.from_numpy(g["fps_start"])
    assert np.array_equal(orc.farthest_point_sample(clouds, 512, st).numpy(), g["fps512"])
    f128 = orc.farthest_point_sample(clouds, 128, st)
    assert np.array_equal(f128.numpy(), g["fps128"])
    new = orc._gather(clouds, f128)
    assert np.array_equal(orc.query_ball_point(0.2, 32, clouds, new).numpy(), g["ball_r02_n32"])
    assert np.array_equal(orc.query_ball_point(0.4, 64, clouds, new).numpy(), g["ball_r04_n64"])
    log = orc.StartLog(util.unpack_starts(g["starts"], g["start_lens"]))
    emb = orc.pointnet2_forward(sd, clouds.permute(0, 2, 1), log, prefix="object_encoder.")
    assert util.rel_l2(emb, g["emb"]) < TOL


def test_dyn3d_forward_and_grads():
    g = util.load("g5_dyn3d.npz")
    B, G, P, L, T, S = g["dims"]
    sd = util.dyn3d_sd(g["seed"])
    f = lambda k: torch.from_numpy(g[k])
    log = orc.StartLog(util.unpack_starts(g["fwd_starts"], g["fwd_start_lens"]))
    y = orc.dyn3d_forward(sd, f("fwd_xc"), f("fwd_xo"), f("fwd_xp"), f("fwd_t"), f("fwd_clouds").permute(0, 2, 1), log)
    assert util.rel_l2(y, g["fwd_logits"]) < TOL
    x, obj = f("x"), f("obj")
    t = torch.full((int(B),), 6, dtype=torch.int64)
    for sub in (7, 512):
        s = util.setup('point_3d', None, sd, T, S, L, G, P, sub)
        for o in ('rotate', 'clockwise_left', 'convergence'):
            log = orc.StartLog(util.unpack_starts(g[f"starts_{o}_sub{sub}"], g[f"start_lens_{o}_sub{sub}"]))
            gr = orc.cond_fn(s, x, t, o, obj, (-1.0, 1.0), centers=torch.tensor([1, 0]) if o == 'convergence' else None,
                             starts=log)
            assert util.rel_l2(gr, g[f"grad_{o}_sub{sub}"]) < 1e-5, (o, sub)
    # same seed -> the oracle's own torch.randint draws reproduce the reference's (pointnet2_utils.py:83)
    s = util.setup('point_3d', None, sd, T, S, L, G, P, 7)
    torch.manual_seed(99)
    log = orc.StartLog()
    orc.cond_fn(s, x, t, 'rotate', obj, (-1.0, 1.0), starts=log)
    assert np.array_equal(torch.cat(log.log).numpy(), g["starts_rotate_sub7"])


def test_full_chains():
    g = util.load("g6_chains.npz")
    unet = util.unet_sd(g["unet_seed"])
    from dgdm_amd import synth
    # 2-D
    B, G, P, L, T, S, nv = g["dims2d"]
    s = util.setup('point', unet, util.dyn2d_sd(g["dyn2d_seed"], nv), T, S, L, G, P)
    noise = synth.synth_noise(0, int(B), int(L))
    ug = orc.unguided_sample(s, noise)
    assert util.rel_l2(ug, g["unguided2d"]) < 1e-5
    objs = torch.from_numpy(g["objs2d"])
    for o in ('rotate', 'shift_left', 'counterclockwise_up', 'convergence'):
        for oi in range(2):
            out = orc.guided_sample(s, noise, objs[oi], o, (-1.0, 1.0), unguided=ug)
            assert util.rel_l2(out, g[f"guided2d_{o}_obj{oi}"]) < 1e-4, (o, oi)
    out = orc.guided_sample_multi_object(s, noise, list(objs), 'rotate_clockwise')
    assert util.rel_l2(out, g["multi2d_rotate_clockwise"]) < 1e-4
    # 3-D
    B, G, P, L, T, S = g["dims3d"]
    s = util.setup('point_3d', unet, util.dyn3d_sd(g["dyn3d_seed"]), T, S, L, G, P, 5)
    noise = synth.synth_noise(0, int(B), int(L))
    ug = orc.unguided_sample(s, noise)
    assert util.rel_l2(ug, g["unguided3d"]) < 1e-5
    objs = torch.from_numpy(g["objs3d"])
    for o in ('rotate', 'convergence'):
        log = orc.StartLog(util.unpack_starts(g[f"guided3d_{o}_starts"], g[f"guided3d_{o}_start_lens"]))
        out = orc.guided_sample(s, noise, objs[0], o, (-1.0, 1.0), unguided=ug, starts=log)
        assert util.rel_l2(out, g[f"guided3d_{o}"]) < 1e-4, o
    log = orc.StartLog(util.unpack_starts(g["multi3d_shift_up_starts"], g["multi3d_shift_up_start_lens"]))
    out = orc.guided_sample_multi_object(s, noise, list(objs), 'shift_up', starts=log)
    assert util.rel_l2(out, g["multi3d_shift_up"]) < 1e-4


def test_oracle_replays_reference_trajectory():
    """Step-by-step: the oracle's eps-net / cond_fn / scheduler step on the reference's recorded per-step inputs."""
    g = util.load("g6_chains.npz")
    unet = util.unet_sd(g["unet_seed"])
    B, G, P, L, T, S, nv = g["dims2d"]
    s = util.setup('point', unet, util.dyn2d_sd(g["dyn2d_seed"], nv), T, S, L, G, P)
    objs = torch.from_numpy(g["objs2d"])
    xs, es, gs = (torch.from_numpy(g["trace2d_rotate_obj1" + k]) for k in ("_x", "_eps", "_grad"))
    for si, t in enumerate(s.sched.timesteps):
        ts = t * torch.ones(int(B), dtype=torch.int64)
        assert util.rel_l2(orc.unet1d_forward(unet, xs[si], ts), es[si]) < TOL
        assert util.rel_l2(orc.cond_fn(s, xs[si], ts, 'rotate', objs[1]), gs[si]) < 1e-5
        nxt = s.sched.step(es[si] - (1 - s.sched.alphas_cumprod[t]).sqrt() * gs[si] * orc.SCALE_2D, t, xs[si])
        want = xs[si + 1] if si + 1 < len(xs) else torch.from_numpy(g["guided2d_rotate_obj1"])
        assert util.rel_l2(nxt, want) < 1e-6


def test_convergence_helpers():
    g = util.load("g7_convergence.npz")
    for k in ("all0", "all2", "all1", "wrap", "mixed", "single", "ones_between"):
        l, c = orc.convergence_mode_three_class(torch.from_numpy(g[f"{k}_profile"]))
        assert np.array_equal(l.numpy(), g[f"{k}_lengths"]) and np.array_equal(c.numpy(), g[f"{k}_centers"]), k
    a = torch.arange(10.0)
    for i in range(6):
        lo, hi = g[f"slicer_{i}_args"]
        assert np.array_equal(orc.slicer(a, int(lo), int(hi)).numpy(), g[f"slicer_{i}"])
    B, G, P, nv = g["cc2d_dims"]
    s = util.setup('point', None, util.dyn2d_sd(g["dyn2d_seed"], nv), 15, 5, 14, G, P)
    c = orc.get_convergence_centers(s, torch.from_numpy(g["cc2d_unguided"]), torch.from_numpy(g["cc2d_obj"]))
    assert np.array_equal(c.numpy(), g["cc2d_centers"])
    B, G, P = g["cc3d_dims"]
    s = util.setup('point_3d', None, util.dyn3d_sd(g["dyn3d_seed"]), 15, 5, 42, G, P, 4)
    log = orc.StartLog(util.unpack_starts(g["cc3d_starts"], g["cc3d_start_lens"]))
    c = orc.get_convergence_centers(s, torch.from_numpy(g["cc3d_unguided"]), torch.from_numpy(g["cc3d_obj"]), starts=log)
    assert np.array_equal(c.numpy(), g["cc3d_centers"])


def _train_case(g, tag, make):
    """Drives a Trainer-like object through the schedule of make_golden.g10_train2d and returns what the fixture recorded."""
    n_g, n_p, L, nv, T = [int(v) for v in g["dims"]]
    wd = {"wd0": 0.0, "wd1": 0.01}[tag]
    tr = make(util.dyn2d_sd(g["dyn2d_seed"], nv), T, wd)
    data = util.train2d_data(int(g["data_seed"]), n_g, n_p)
    torch.manual_seed(int(g["torch_seed"]))
    rec = {}
    for step in range(3):
        if step == 2:
            tr.lr_step()
        loss, pred = tr.step(*data)
        rec[f"loss{step}"], rec[f"pred{step}"] = loss, pred
        if step == 0:
            rec["grads"] = tr.gradients()
    rec["final"] = tr.state_dict()
    rec["inf_pred"], rec["inf_loss"] = tr.inference(*data)
    rec["trainer"] = tr
    return rec


class _OracleTrainer:
    def __init__(self, sd, T, wd):
        self.t = orc.Trainer2D(sd, T, 1e-4, wd)
        self.epoch = 0

    def lr_step(self):      # CosineAnnealingLR(T_max=100, eta_min=1e-2 * lr) (dynamics/trainer.py:47)
        self.epoch += 1
        self.t.lr = 1e-6 + (1e-4 - 1e-6) * (1 + np.cos(np.pi * self.epoch / 100)) / 2

    def step(self, *a):
        return self.t.step(*a)

    def inference(self, *a):
        return self.t.inference(*a)

    def gradients(self):
        return self.t.grads

    def state_dict(self):
        return self.t.sd


# Biases that only shift the input of a BatchNorm by the same vector in every row - the Linear biases in front of each BatchNorm
# and, through linears.0, the output biases of the three encoders: their gradient is zero in exact arithmetic (the batch mean
# absorbs the shift), what any implementation computes is rounding noise of column sums, and Adam turns noise of any size into
# steps of +-lr.  They are compared nowhere; every other tensor is.
BN_FED_BIAS = {f"linears.{3 * i}.bias" for i in range(8)} | {f"{e}_encoder.2.bias" for e in ("gripper", "object", "time")}


def check_training(g, tag, rec, tol_pred, tol_grad, tol_param):
    for step in range(3):
        assert abs(rec[f"loss{step}"] / float(g[f"{tag}_loss{step}"]) - 1) < tol_pred, (tag, step)
        assert util.rel_l2(rec[f"pred{step}"].cpu(), g[f"{tag}_pred{step}"]) < tol_pred, (tag, step)
    worst = {}
    for key in [k for k in g.files if k.startswith(f"{tag}_grad/")]:
        name = key.split("/", 1)[1]
        if name in BN_FED_BIAS:
            continue
        mine = rec["grads"][name].double().flatten()
        ref = torch.from_numpy(g[key]).double()
        scale = float(np.sqrt(g[f"{tag}_gradsum/{name}"][1] / mine.numel()))          # rms of the reference's gradient tensor
        worst[name] = float((mine[util.sample_idx(name, mine.numel())] - ref).abs().max()) / scale
        assert worst[name] < tol_grad, (tag, name, worst[name])
        assert abs(float((mine * mine).sum()) / g[f"{tag}_gradsum/{name}"][1] - 1) < 10 * tol_grad, (tag, name)
    for key in [k for k in g.files if k.startswith(f"{tag}_final/")]:
        name = key.split("/", 1)[1]
        if name in BN_FED_BIAS:
            continue
        mine = rec["final"][name].double().flatten()
        ref = torch.from_numpy(g[key]).double().flatten()
        if "running_" in name or name.endswith("num_batches_tracked"):
            # a running mean carries the history of the noise-driven biases in front of its BatchNorm (for linears.1 also the three
            # encoder output biases through linears.0's 768 weights per row): 1e-3 / 1e-4 when those can differ, else 1e-5
            tol = 1e-5 if "running_var" in name or tol_param < 3e-6 else (1e-3 if name == "linears.1.running_mean" else 1e-4)
            assert float((mine - ref).abs().max()) < tol, (tag, name, float((mine - ref).abs().max()))
        else:
            # Three Adam steps move every entry by about 3 lr = 3e-4, whatever the size of its gradient (Adam normalises): the
            # tolerance is a small fraction of that.  Entries whose gradient is rounding residue (below 1e-3 of the tensor's
            # largest: e.g. the bias of an encoder unit that is active for every object of the batch, whose gradient is a column
            # sum BatchNorm makes zero) take noise-driven steps; for those only the step bound is checked.
            d = (mine[util.sample_idx(name, mine.numel())] - ref).abs()
            gref = torch.from_numpy(g[f"{tag}_grad/{name}"]).double().abs()
            real = gref >= 1e-3 * gref.max()
            assert float(d[real].max()) < tol_param, (tag, name, float(d[real].max()))
            assert float(d.max()) < 2 * 3.1e-4, (tag, name)
    # eval mode sees the noise-driven biases against running means that averaged over their earlier values (each up to 3 lr apart
    # between two implementations): 2e-3, not tol_pred.  (tests/test_gpu_train.py checks the eval forward tightly, on its own weights.)
    assert abs(rec["inf_loss"] / float(g[f"{tag}_inf_loss"]) - 1) < 2e-3
    assert util.rel_l2(rec["inf_pred"].cpu(), g[f"{tag}_inf_pred"]) < 2e-3
    return worst


@pytest.mark.parametrize("tag", ["wd0", "wd1"])
def test_trainer2d_matches_reference(tag):
    """oracle.Trainer2D against the reference's own Trainer.step / inference (three steps, one schedule step, then eval)."""
    g = util.load("g10_train2d.npz")
    rec = _train_case(g, tag, _OracleTrainer)
    check_training(g, tag, rec, 5e-6, 2e-5, 2e-6)


@pytest.mark.parametrize("tag", ["p2", "p3"])
def test_unet_trainer_matches_reference(tag):
    """oracle.UnetTrainer against the reference's own Diffusion.get_stats + torch.optim.Adam (make_golden.g12_unet_train): three steps
    on the recorded draws - losses, the gradients of step 1, the parameters after step 3."""
    import math
    from tests.util import sample_idx
    g = util.load("g12_unet_train.npz")
    B, L = [int(v) for v in g[f"{tag}_dims"]]
    lr0, E = float(g["lr"]), int(g["num_epochs"])
    o = orc.UnetTrainer(util.unet_sd(int(g["unet_seed"])), int(g["num_train_timesteps"]), L, lr0, ema_power=0.85)
    x0 = torch.from_numpy(g[f"{tag}_x0"])
    torch.manual_seed(int(g["torch_seed"]))
    for step in range(3):
        if step == 2:
            o.lr = lr0 * (1 + math.cos(math.pi / E)) / 2
        loss, _ = o.step(x0)                 # draws torch.randn / torch.randint itself, in get_stats' order
        assert np.array_equal(o.draws[0].numpy(), g[f"{tag}_noise{step}"]) and np.array_equal(o.draws[1].numpy(), g[f"{tag}_t{step}"])
        assert abs(loss - float(g[f"{tag}_loss{step}"])) < 1e-6
        if step == 0:
            for k, v in o.grads.items():
                f = v.double().flatten().numpy()
                rms = math.sqrt(float(g[f"{tag}_gradsum/{k}"][1]) / f.size)
                assert np.abs(f[sample_idx(k, f.size)] - g[f"{tag}_grad/{k}"]).max() < 2e-5 * rms, k
    for k, v in o.sd.items():
        f = v.double().flatten().numpy()
        assert np.abs(f[sample_idx(k, f.size)] - g[f"{tag}_final/{k}"]).max() < 4e-6, k
    # EMAModel's schedule (diffusers 0.11.1, restated: parity unpinned): no averaging on the first two calls, then 1 - (1 + step)^-power
    assert o.ema.optimization_step == 3 and abs(o.ema.decay - (1 - 2 ** -0.85)) < 1e-12


@pytest.mark.parametrize("tag", ["plain", "sub"])
def test_trainer3d_matches_reference(tag):
    """oracle.Trainer3D against the reference's own Trainer.step / inference with --fingers_3d (PointNet++ in training mode; with and
    without --use_sub_batch)."""
    from tests import train3d_common as t3
    g = util.load("g13_train3d.npz")
    # (running statistics after three steps: 2e-3 - Adam moves every entry whose gradient is rounding residue by +-lr per step, different
    # entries in two float32 evaluations, and BatchNorm over 8 rows passes that on to the later forwards (1e-2) and to the eval-mode
    # inference after the steps (2e-2), the bounds the HIP test uses; the first step is held to 2e-5 / 5e-5 / 4e-6)
    rec = t3.drive(g, tag, t3.OracleTrainer3D)
    if tag == "sub":
        t3.check_sub_loosely(g, rec)        # (two optimizer steps per call with BatchNorm over 4 rows in between: see there)
    else:
        t3.check(g, tag, rec, 2e-5, 5e-5, 4e-6, verbose=True, tol_run=2e-3, tol_later=1e-2, tol_inf=2e-2)
