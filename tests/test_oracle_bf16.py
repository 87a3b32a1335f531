"""CPU checks of the oracle's bf16-contraction mode (the statement trunk_bf16.hip is tested against on the GPU)."""
import torch

from dgdm_amd import synth
from oracle import dgdm_oracle as orc
from tests import util


def test_bf_linear_matches_explicit_rounding():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 64, generator=g, requires_grad=True)
    w = torch.randn(32, 64, generator=g)
    y = orc._BfLinear.apply(x, w)
    xb, wb = x.detach().bfloat16().float(), w.bfloat16().float()
    assert torch.equal(y, xb @ wb.t())
    gy = torch.randn(7, 32, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    assert torch.equal(gx, gy.bfloat16().float() @ wb)


def test_bf16_mode_is_scoped_and_close_to_f32():
    B, G, P, L, T, nv = 2, 6, 2, 14, 15, 100
    sd = util.dyn2d_sd(22, nv)
    s = util.setup('point', None, sd, T, 5, L, G, P)
    x = synth.synth_noise(5, B, L).clamp(-1, 1)
    ts = torch.full((B,), 6, dtype=torch.int64)
    obj = synth.synth_object_2d(0, nv)
    a = orc.cond_fn(s, x, ts, 'shift_left', obj)
    with orc.contraction('bf16'):
        b = orc.cond_fn(s, x, ts, 'shift_left', obj)
    assert torch.equal(orc.cond_fn(s, x, ts, 'shift_left', obj), a)          # the switch does not leak
    e = util.rel_l2(b, a)
    assert 1e-4 < e < 1e-1, e                                               # bf16 operands: a percent-level change, not noise, not garbage


def test_bf16_trunk_folding_matches_batchnorm():
    """With the rounding removed, the folded trunk of the bf16 statement equals the Linear->BatchNorm->ReLU trunk."""
    sd = util.dyn2d_sd(3, 100)
    x = torch.randn(9, 3 * 256 + 27, generator=torch.Generator().manual_seed(1))
    ref = orc._trunk(sd, x)
    keep = orc._bf
    try:
        orc._bf = lambda t: t
        with orc.contraction('bf16'):
            got = orc._trunk(sd, x)
    finally:
        orc._bf = keep
    assert util.rel_l2(got, ref) < 1e-5
