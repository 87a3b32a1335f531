"""GPU tests at BASELINE.json's full sizes, through properties that do not need the (hours-long) CPU oracle at that size:
linearity / antisymmetry of the guidance gradient in the objective, independence of chains that share a launch,
equality of the two FPS paths, and per-row spot checks of the forward sweep against the oracle."""
import numpy as np
import pytest
import torch

from dgdm_amd import engine, sampler, synth
from oracle import dgdm_oracle as orc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from dgdm_amd import _lib
    _lib.device_init(0)
    return torch.device("cuda:0")


def test_2d_full_size_properties(dev):
    B, G, P, L, nv, T = 64, 360, 5, 14, 100, 15                       # BASELINE configs[1]: R = 576 000 rows per cond_fn
    sd = util.dyn2d_sd(22, nv)
    dyn = engine.Dynamics(2, sd, L, 2 * nv)
    objs = torch.stack([synth.synth_object_2d(i, nv) for i in range(2)])
    gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 4, T, nv, 0, max_objects=2)
    gd.set_objects(objs.to(dev))
    assert gd.rows == 576000
    x = synth.synth_noise(1, B, L).clamp(-1, 1).reshape(1, B, L).to(dev)
    mk = engine.make_objective
    names = ['rotate_clockwise', 'rotate_counterclockwise', 'shift_up', 'clockwise_up']
    g4 = gd.grad(x.expand(4, -1, -1).contiguous(), 6, [mk(n, 0) for n in names])
    cw, ccw, up, cwup = g4
    # antisymmetry in the objective: to rounding (the bf16 matrix pipe that carries the split-float32 contractions does not negate bit
    # for bit), and bit for bit on the float32 MFMA chain
    assert util.rel_l2(cw.cpu(), (-ccw).cpu()) < 2e-6
    gd.set_contraction_dtype("f32_mfma")
    m2 = gd.grad(x.expand(2, -1, -1).contiguous(), 6, [mk(n, 0) for n in names[:2]])
    assert torch.equal(m2[0], -m2[1]) and util.rel_l2(m2[0].cpu(), cw.cpu()) < 1e-4      # the two float32 forms agree to the rounding noise of a 9000-cell sum
    gd.set_contraction_dtype("f32")
    assert util.rel_l2(cwup.cpu(), (cw + up).cpu()) < 2e-6             # objective is linear in the deltas
    alone = gd.grad(x, 6, [mk('shift_up', 0)])                         # a chain does not see its launch neighbours
    assert torch.equal(alone[0], up)
    other = gd.grad(x.expand(2, -1, -1).contiguous(), 6, [mk('shift_up', 1), mk('shift_up', 0)])
    assert torch.equal(other[1], up) and not torch.equal(other[0], up)
    assert bool(torch.isfinite(g4).all()) and float(cw.abs().max()) > 0
    # spot check of the forward path at this size: orientation sweep rows vs the oracle's model forward
    xs = x[0].cpu()
    logits = gd.sweep(x, [1]).cpu()[0]                                 # (B*G, 3), row = g*B + b
    rs = np.random.RandomState(0)
    rows = rs.choice(B * G, size=64, replace=False)
    ori = torch.linspace(-1.0, 1.0, G)
    bsel, gsel = rows % B, rows // B
    ref = orc.dyn2d_forward(sd, xs[bsel], ori[gsel].reshape(-1, 1), torch.zeros(len(rows), 2), torch.zeros(len(rows)),
                            objs[1].reshape(1, -1).expand(len(rows), -1))
    assert util.rel_l2(logits[rows], ref) < 2e-5


def test_3d_full_size_properties(dev):
    B, G, P, L, T, sub = 32, 45, 5, 42, 15, 512                         # BASELINE configs[2]: R = 36 000, 71 sub-batches
    sd = util.dyn3d_sd(33)
    dyn = engine.Dynamics(3, sd, L)
    objs = torch.stack([synth.synth_object_3d(70 + i) for i in range(2)])
    gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 3, T, 512, sub, max_objects=2)
    gd.set_objects(objs.to(dev))
    assert gd.rows == 36000 and gd.starts_per_call == 72000
    x = synth.synth_noise(2, B, L).clamp(-1, 1).reshape(1, B, L).to(dev)
    torch.manual_seed(7)
    st = sampler.StartStream(512, sub).call(gd.rows)
    mk = engine.make_objective
    names = ['shift_left', 'shift_right', 'rotate']
    g3 = gd.grad(x.expand(3, -1, -1).contiguous(), 3, [mk(n, 0) for n in names], None, np.concatenate([st, st, st]))
    assert util.rel_l2(g3[0].cpu(), (-g3[1]).cpu()) < 2e-6           # see test_2d_full_size_properties
    gd.set_contraction_dtype("f32_mfma")
    m2 = gd.grad(x.expand(2, -1, -1).contiguous(), 3, [mk(n, 0) for n in names[:2]], None, np.concatenate([st, st]))
    assert torch.equal(m2[0], -m2[1])
    tie = util.finger_err(m2[0].cpu().reshape(B, L, 1), g3[0].cpu().reshape(B, L, 1)) / m2[0].cpu().reshape(B, -1).norm(dim=1).double()
    assert float(tie.median()) < 2e-6 and float(tie.max()) < 1e-3, tie     # the two float32 forms: rounding, bar a ReLU tie in a finger
    gd.set_contraction_dtype("f32")
    gd.debug_fps_path(True)                                            # every row runs its own FPS(128)
    slow = gd.grad(x, 3, [mk('rotate', 0)], None, st)
    gd.debug_fps_path(False)
    assert torch.equal(slow[0], g3[2])
    assert bool(torch.isfinite(g3).all()) and float(g3[2].abs().max()) > 0
    # forward spot check: orientation sweep rows (sub-batch partition of get_convergence_centers) vs the oracle
    torch.manual_seed(8)
    sw = sampler.StartStream(512, sub).call(gd.sweep_rows)
    logits = gd.sweep(x, [1], sw).cpu()[0]
    rows = np.random.RandomState(1).choice(B * G, size=12, replace=False)
    s1 = np.empty(B * G, np.int64)
    s2 = np.empty(B * G, np.int64)
    for r0 in range(0, B * G, sub):
        n = min(sub, B * G - r0)
        s1[r0:r0 + n], s2[r0:r0 + n] = sw[2 * r0:2 * r0 + n], sw[2 * r0 + n:2 * r0 + 2 * n]
    ori = torch.linspace(-1.0, 1.0, G)
    xs = x[0].cpu()
    bsel, gsel = rows % B, rows // B
    lin = torch.linspace(-1.0, 1.0, L // 2).repeat(2).reshape(1, 1, -1).expand(len(rows), -1, -1)
    pts = torch.cat([lin, xs[bsel].reshape(len(rows), 1, L), lin], dim=1)
    log = orc.StartLog([torch.from_numpy(s1[rows]), torch.from_numpy(s2[rows])])
    ref = orc.dyn3d_forward(sd, pts, ori[gsel].reshape(-1, 1), torch.zeros(len(rows), 2), torch.zeros(len(rows)),
                            objs[1].t().unsqueeze(0).expand(len(rows), -1, -1), log)
    assert util.rel_l2(logits[rows], ref) < 2e-5
