"""Dynamic range of the default float32-grade arithmetic (three f16 MFMA products per float32 product, csrc/trunk_f16l.hip and
csrc/unet.hip conv_mfma_f16x3) on ADVERSARIALLY scaled checkpoints.

f16 has five exponent bits, so both operands are scaled by exact powers of two before they are split; the weight side's scale is fixed
on the host.  The synthetic checkpoints of the other tests (He-init weights, running_var ~ U(0.5, 1.5)) never stress that choice - a
released checkpoint may: BatchNorm-folded per-channel gains spanning many binades inside one matrix, dead channels, running variances
from 1e-6 to 1e2.  The checkpoints built here have all of that:

  dynamics   every hidden unit j of every trunk layer gets a gain s_j = 2^U(-12, 12) on its BatchNorm affine output (weight and bias;
             ReLU commutes with it) and 1 / s_j on its column of the next layer - the function is unchanged, the weights of ONE matrix now
             span 2^24 per row and per column; 5 % of the units are dead (gamma = beta = 0); running_var ~ 10^U(-6, 2) with gamma
             rescaled to keep the unit's size.  `compensate=False`: gains 2^U(-24, 0) with NO inverse downstream (a net whose units really
             differ by that much).
  eps-net    every GroupNorm group of every convolution that feeds a GroupNorm gets a gain 2^U(-12, 12) on its weights and bias (the
             normalisation absorbs it up to its eps); 5 % of the output channels of every convolution are dead (zero row, zero bias).

Yardstick: the CPU oracle evaluated in float64 on the same float32 checkpoint (oracle/dgdm_oracle.py; float64 weights make the whole
model run in float64).  Asserted: the default form is finite everywhere and within max(1e-6, 2 x err(float32 MFMA chain)) of float64 -
i.e. float32-grade wherever the float32 chain itself is.  What makes it hold (both added with this test): the trunk is equilibrated on
the host by exact powers of two per hidden unit (models_api.hip TrunkEquil; the float32 forms are bit-identical under it), the eps-net's
f16 images carry one scale per OUTPUT CHANNEL.  `DGDM_NO_EQUILIBRATION=1` shows the trunk without: reported, and asserted to be worse on
the compensated checkpoint (the test has teeth)."""
import os

import numpy as np
import pytest
import torch

from dgdm_amd import engine, sampler, synth
from oracle import dgdm_oracle as orc
from tests import util
from tests.test_gpu_parity import dev      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
FLOOR = 1e-6


def adversarial_dyn_sd(sd, seed, compensate=True, dead=0.05, bias="scaled"):
    """See the module docstring.  Trunk layers only (linears.{3l} + BatchNorm linears.{3l+1}, l = 0 .. 7, then `output`).
    bias='scaled': the gain on gamma and beta together (the function is unchanged when compensated downstream); 'kept': on gamma ALONE
    (beta as it was: another network, whose units' weights span 2^24 while their biases - and so their activations - do not);
    'near_dead': 10 % of the units with gamma x 1e-12 and beta ~ 0.1 (a BatchNorm channel that has died in training but keeps its
    offset: its activation is an ordinary constant, its weight row is 40 binades below its neighbours')."""
    rs = np.random.RandomState(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    for l in range(8):
        lin, bn = f"linears.{3 * l}", f"linears.{3 * l + 1}"
        n = sd[bn + ".weight"].shape[0]
        if bias == "near_dead":
            nd = rs.uniform(size=n) < 0.10
            sd[bn + ".weight"] = (sd[bn + ".weight"].double() * torch.from_numpy(np.where(nd, 1e-12, 1.0))).float()
            sd[bn + ".bias"] = torch.from_numpy(np.where(nd, 0.1 * rs.uniform(0.5, 1.5, n), sd[bn + ".bias"].double().numpy())).float()
            continue
        s = 2.0 ** (rs.uniform(-12, 12, n) if compensate else rs.uniform(-24, 0, n))
        kill = rs.uniform(size=n) < dead
        var_new = 10.0 ** rs.uniform(-6, 2, n)
        var_old = sd[bn + ".running_var"].double().numpy()
        keep = np.sqrt((var_new + 1e-5) / (var_old + 1e-5))                      # gamma / sqrt(var + eps) as before
        gain = torch.from_numpy(np.where(kill, 0.0, s * keep))
        sd[bn + ".running_var"] = torch.from_numpy(var_new).float()
        sd[bn + ".weight"] = (sd[bn + ".weight"].double() * gain).float()
        sd[bn + ".bias"] = (sd[bn + ".bias"].double() * torch.from_numpy(np.where(kill, 0.0, s if bias == "scaled" else 1.0))).float()
        if compensate:
            nxt = f"linears.{3 * (l + 1)}.weight" if l < 7 else "output.weight"
            sd[nxt] = (sd[nxt].double() / torch.from_numpy(s)[None, :]).float()
    return sd


def adversarial_unet_sd(sd, seed, n_groups=8, dead=0.05):
    rs = np.random.RandomState(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    for k in [k for k in sd if k.endswith(".block.0.weight")]:
        w, b = sd[k], sd[k[:-6] + "bias"]
        cout = w.shape[0]
        g = np.repeat(2.0 ** rs.uniform(-12, 12, n_groups), cout // n_groups)
        g[rs.uniform(size=cout) < dead] = 0.0
        g = torch.from_numpy(g)
        sd[k] = (w.double() * g[:, None, None]).float()
        sd[k[:-6] + "bias"] = (b.double() * g).float()
    return sd


def _grad_err(out, ref):
    """(relative L2 of the whole gradient, the same without the worst finger): a ReLU within rounding of zero moves ONE finger."""
    e = util.finger_err(out, ref).sort().values
    n = float(ref.double().norm())
    return float(e.norm()) / n, float(e[:-1].norm()) / n


def _check(tag, errs):
    """errs: mode -> (all fingers, all but the worst).  The default form against the float32 chain's own distance from float64."""
    tol = max(FLOOR, 2.0 * errs["f32_mfma"][0])
    print(f"{tag}: vs float64 - f16x3 {errs['f32'][0]:.2e}, float32 MFMA chain {errs['f32_mfma'][0]:.2e}"
          + (f", f16x3 without equilibration {errs['noeq'][0]:.2e}" if "noeq" in errs else ""))
    # one finger may hold a ReLU tie that falls differently than in float64 (tests/test_gpu_fullgrid.py): then every OTHER finger is held to the bound
    assert errs["f32"][0] < tol or (errs["f32"][1] < tol and errs["f32"][0] < 1e-3), (tag, errs)


def _dyn(kind, sd, L, nv=0, equilibrate=True):
    if not equilibrate:
        os.environ["DGDM_NO_EQUILIBRATION"] = "1"
    try:
        return engine.Dynamics(kind, sd, L, 2 * nv) if kind == 2 else engine.Dynamics(3, sd, L)
    finally:
        os.environ.pop("DGDM_NO_EQUILIBRATION", None)


@pytest.mark.parametrize("compensate", [True, False])
def test_f16x3_dynamic_range_2d(dev, compensate):      # noqa: F811
    nv, B, G, P, L, T = 100, 4, 8, 3, 14, 15
    sd = adversarial_dyn_sd(util.dyn2d_sd(91, nv), 5, compensate)
    sd64 = {k: v.double() for k, v in sd.items()}
    obj = synth.synth_object_2d(3, nv)
    x = synth.synth_noise(70, B, L).clamp(-1, 1)
    s64 = util.setup('point', None, sd64, T, 5, L, G, P)
    for o in ("rotate", "shift_left", "counterclockwise_up"):
        ref = orc.cond_fn(s64, x.double(), torch.full((B,), 6, dtype=torch.int64), o, obj.double())
        assert bool(torch.isfinite(ref).all()) and float(ref.norm()) > 0
        errs = {}
        for tag, mode, eq in (("f32", "f32", True), ("f32_mfma", "f32_mfma", True), ("noeq", "f32", False)):
            gd = engine.Guidance(_dyn(2, sd, L, nv, eq), B, G, P, (-1.0, 1.0), 1, T, nv, 0, max_objects=1, contraction_dtype=mode)
            gd.set_objects(obj[None].to(dev))
            gr = gd.grad(x.reshape(1, B, L).to(dev), 6, [engine.make_objective(o, 0)], None).cpu()
            assert bool(torch.isfinite(gr).all()), (tag, o)
            errs[tag] = _grad_err(gr.reshape(B, L, 1), ref)
        _check(f"2-D {o} compensate={compensate}", errs)
        if compensate:
            assert errs["noeq"][0] > errs["f32"][0], errs         # without the host-side equilibration the one-scale-per-matrix split degrades


def _starts_3d(rows, sub, seed):
    g = torch.Generator().manual_seed(seed)
    calls = []
    for r0 in range(0, rows, sub):
        n = min(sub, rows - r0)
        calls += [torch.randint(0, 512, (n,), generator=g), torch.randint(0, 512, (n,), generator=g)]
    return calls


def _grad_3d(dev, sd, obj, x, t, o, B, G, P, L, T, sub, calls, mode, equilibrate=True):
    gd = engine.Guidance(_dyn(3, sd, L, 0, equilibrate), B, G, P, (-1.0, 1.0), 1, T, 512, sub, max_objects=1, contraction_dtype=mode)
    gd.set_objects(obj[None].to(dev))
    st = sampler.StartStream(512, sub, [c.clone() for c in calls])
    return gd.grad(x.reshape(1, B, L).to(dev), t, [engine.make_objective(o, 0)], None, st.call(gd.rows)).cpu()


@pytest.mark.parametrize("compensate", [True, False])
def test_f16x3_dynamic_range_3d(dev, compensate):      # noqa: F811
    B, G, P, L, T, sub = 2, 3, 2, 42, 15, 7
    sd = adversarial_dyn_sd(util.dyn3d_sd(92), 6, compensate)
    sd64 = {k: v.double() for k, v in sd.items()}
    obj = synth.synth_object_3d(53)
    x = synth.synth_noise(71, B, L).clamp(-1, 1)
    calls = _starts_3d(B * G * P * P, sub, 8)
    s64 = util.setup('point_3d', None, sd64, T, 5, L, G, P, sub)
    for o in ("rotate", "shift_up"):
        ref = orc.cond_fn(s64, x.double(), torch.full((B,), 9, dtype=torch.int64), o, obj.double(), (-1.0, 1.0), None, orc.StartLog([c.clone() for c in calls]))
        assert bool(torch.isfinite(ref).all()) and float(ref.norm()) > 0
        errs = {}
        for tag, mode, eq in (("f32", "f32", True), ("f32_mfma", "f32_mfma", True), ("noeq", "f32", False)):
            gr = _grad_3d(dev, sd, obj, x, 9, o, B, G, P, L, T, sub, calls, mode, eq)
            assert bool(torch.isfinite(gr).all()), (tag, o)
            errs[tag] = _grad_err(gr.reshape(B, L, 1), ref)
        _check(f"3-D {o} compensate={compensate}", errs)
        if compensate:
            assert errs["noeq"][0] > errs["f32"][0], errs


@pytest.mark.parametrize("bias", ["kept", "near_dead"])
def test_f16x3_bias_dominated_units(dev, bias):      # noqa: F811
    """Units whose WEIGHT row is tiny against their bias (round-5 advisor): gains on gamma alone (beta unchanged), and near-dead BatchNorm
    channels (gamma ~ 1e-12, beta ~ 0.1).  The equilibration must not lift such a row's weights - its activation, an ordinary
    constant, would become 2^24 .. 2^60 times its neighbours' and take the tile row's f16 scale with it (models_api.hip TrunkEquil counts
    the bias as a weight on the layer's input scale).  2-D and 3-D, against float64."""
    T = 15
    nv, B2, G2, P2, L2 = 100, 4, 8, 3, 14
    sd = adversarial_dyn_sd(util.dyn2d_sd(97, nv), 12, True, bias=bias)
    sd64 = {k: v.double() for k, v in sd.items()}
    obj = synth.synth_object_2d(5, nv)
    x = synth.synth_noise(76, B2, L2).clamp(-1, 1)
    s64 = util.setup('point', None, sd64, T, 5, L2, G2, P2)
    for o in ("rotate", "shift_left"):
        ref = orc.cond_fn(s64, x.double(), torch.full((B2,), 6, dtype=torch.int64), o, obj.double())
        assert bool(torch.isfinite(ref).all()) and float(ref.norm()) > 0
        errs = {}
        for tag, mode in (("f32", "f32"), ("f32_mfma", "f32_mfma")):
            gd = engine.Guidance(_dyn(2, sd, L2, nv), B2, G2, P2, (-1.0, 1.0), 1, T, nv, 0, max_objects=1, contraction_dtype=mode)
            gd.set_objects(obj[None].to(dev))
            gr = gd.grad(x.reshape(1, B2, L2).to(dev), 6, [engine.make_objective(o, 0)], None).cpu()
            assert bool(torch.isfinite(gr).all()), (tag, o)
            errs[tag] = _grad_err(gr.reshape(B2, L2, 1), ref)
        _check(f"2-D {o} bias={bias}", errs)
    B, G, P, L, sub = 2, 3, 2, 42, 7
    sd = adversarial_dyn_sd(util.dyn3d_sd(98), 13, True, bias=bias)
    sd64 = {k: v.double() for k, v in sd.items()}
    obj = synth.synth_object_3d(56)
    x = synth.synth_noise(77, B, L).clamp(-1, 1)
    calls = _starts_3d(B * G * P * P, sub, 11)
    s64 = util.setup('point_3d', None, sd64, T, 5, L, G, P, sub)
    for o in ("rotate", "shift_up"):
        ref = orc.cond_fn(s64, x.double(), torch.full((B,), 9, dtype=torch.int64), o, obj.double(), (-1.0, 1.0), None, orc.StartLog([c.clone() for c in calls]))
        assert bool(torch.isfinite(ref).all()) and float(ref.norm()) > 0
        errs = {}
        for tag, mode in (("f32", "f32"), ("f32_mfma", "f32_mfma")):
            gr = _grad_3d(dev, sd, obj, x, 9, o, B, G, P, L, T, sub, calls, mode)
            assert bool(torch.isfinite(gr).all()), (tag, o)
            errs[tag] = _grad_err(gr.reshape(B, L, 1), ref)
        _check(f"3-D {o} bias={bias}", errs)


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 1e3, 1e6])
def test_f16x3_layer1_bound_sweep_3d(dev, scale):      # noqa: F811
    """scripts/trunk_range.py as a test: 3-D layer 1's output - whose f16 row scale for layer 2 comes from a BOUND, not from the values
    (trunk_f16l.hip header) - moved over twelve orders of magnitude (first layer and its BatchNorm statistics x s, layer 2's input
    weights / s: the same function), against float64."""
    B, G, P, L, T, sub = 2, 3, 2, 42, 15, 7
    sd = {k: v.clone() for k, v in util.dyn3d_sd(93).items()}
    for k in ("linears.1.weight", "linears.1.bias"):          # BatchNorm's affine output x s (its input statistics are untouched)
        sd[k] = sd[k] * scale
    sd["linears.3.weight"] = sd["linears.3.weight"] / scale
    sd64 = {k: v.double() for k, v in sd.items()}
    obj = synth.synth_object_3d(54)
    x = synth.synth_noise(72, B, L).clamp(-1, 1)
    calls = _starts_3d(B * G * P * P, sub, 9)
    ref = orc.cond_fn(util.setup('point_3d', None, sd64, T, 5, L, G, P, sub), x.double(), torch.full((B,), 3, dtype=torch.int64), 'rotate', obj.double(),
                      (-1.0, 1.0), None, orc.StartLog([c.clone() for c in calls]))
    errs = {}
    for tag, mode, eq in (("f32", "f32", True), ("f32_mfma", "f32_mfma", True), ("noeq", "f32", False)):
        gr = _grad_3d(dev, sd, obj, x, 3, 'rotate', B, G, P, L, T, sub, calls, mode, eq)
        assert bool(torch.isfinite(gr).all()), tag
        errs[tag] = _grad_err(gr.reshape(B, L, 1), ref)
    _check(f"3-D layer-1 scale {scale:g}", errs)


def test_equilibration_is_exact_for_float32_forms(dev):      # noqa: F811
    """The host-side equilibration multiplies by powers of two only: the float32 MFMA chain and the bf16-operand trunk return the SAME
    bits with and without it (so every golden test of those forms is a test of the scaled fold as well)."""
    B, G, P, L, T, sub = 2, 3, 2, 42, 15, 7
    sd = adversarial_dyn_sd(util.dyn3d_sd(94), 7, True)
    obj = synth.synth_object_3d(55)
    x = synth.synth_noise(73, B, L).clamp(-1, 1)
    calls = _starts_3d(B * G * P * P, sub, 10)
    for mode in ("f32_mfma", "bf16"):
        a = _grad_3d(dev, sd, obj, x, 6, 'clockwise_left', B, G, P, L, T, sub, calls, mode, True)
        b = _grad_3d(dev, sd, obj, x, 6, 'clockwise_left', B, G, P, L, T, sub, calls, mode, False)
        assert torch.equal(a, b), mode
    nv = 100
    sd2 = adversarial_dyn_sd(util.dyn2d_sd(95, nv), 8, True)
    o2 = synth.synth_object_2d(4, nv)
    x2 = synth.synth_noise(74, 3, 14).clamp(-1, 1)
    for mode in ("f32_mfma", "bf16"):
        res = []
        for eq in (True, False):
            gd = engine.Guidance(_dyn(2, sd2, 14, nv, eq), 3, 8, 3, (-1.0, 1.0), 1, T, nv, 0, max_objects=1, contraction_dtype=mode)
            gd.set_objects(o2[None].to(dev))
            res.append(gd.grad(x2.reshape(1, 3, 14).to(dev), 6, [engine.make_objective('rotate', 0)], None).cpu())
        assert torch.equal(res[0], res[1]), mode


@pytest.mark.parametrize("L", [14, 42])
def test_f16x3_dynamic_range_unet(dev, L):      # noqa: F811
    B = 6
    sd = adversarial_unet_sd(util.unet_sd(96), 11)
    sd64 = {k: v.double() for k, v in sd.items()}
    x = synth.synth_noise(75, B, L)
    t = torch.tensor([0, 3, 12, 14, 7, 1], dtype=torch.int64)
    ref = orc.unet1d_forward(sd64, x.double(), t)
    assert bool(torch.isfinite(ref).all())
    err = {}
    for mode in ("f32", "f32_mfma"):
        out = engine.Unet1d(sd, contraction_dtype=mode).forward(x.to(dev), t.to(dev)).cpu()
        assert bool(torch.isfinite(out).all()), mode
        err[mode] = util.rel_l2(out, ref)
    print(f"eps-net L={L}: vs float64 - f16x3 {err['f32']:.2e}, float32 MFMA chain {err['f32_mfma']:.2e}")
    assert err["f32"] < max(FLOOR, 2.0 * err["f32_mfma"]), err
    # tiny inputs (the input scale and its inverse must stay exact inverses when the exponent clamps)
    tiny = x * 1e-36
    ref_t = orc.unet1d_forward(sd64, tiny.double(), t)
    out_t = engine.Unet1d(sd).forward(tiny.to(dev), t.to(dev)).cpu()
    assert util.rel_l2(out_t, ref_t) < 1e-5
