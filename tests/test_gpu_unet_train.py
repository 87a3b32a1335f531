"""Training of the eps-net on the HIP path (csrc/unet_train.hip) against
* tests/golden/g12_unet_train.npz - the REFERENCE's own Diffusion.get_stats + torch.optim.Adam, three steps (make_golden.py g12), and
* the oracle (oracle.UnetTrainer: torch CPU autograd, pinned to the same fixture by tests/test_oracle_golden.py) at other batch sizes.
Every call goes through the C-ABI (dgdm_unet_trainer_*)."""
import math
import os

import numpy as np
import pytest
import torch

from tests.util import sample_idx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _coefs(T, ts):
    from dgdm_amd.scheduler import DDIMScheduler
    ac = DDIMScheduler(num_train_timesteps=T).alphas_cumprod[ts]
    return ac ** 0.5, (1 - ac) ** 0.5


@pytest.mark.parametrize("tag", ["p2", "p3"])
def test_unet_trainer_golden(tag):
    """Three optimisation steps against the reference's own run: losses 2e-5, every sampled gradient entry of step 1 within 5e-5 of its
    tensor's rms and the tensor's sum of squares to 1e-4, every sampled parameter after three Adam steps within 4e-6 (an Adam step is
    lr * sign-like: a gradient entry whose sign is decided by rounding moves its parameter by 2 lr - counted, at most 2 per case)."""
    from dgdm_amd import engine, synth
    g = np.load(os.path.join(GOLD, "g12_unet_train.npz"))
    B, L = [int(v) for v in g[f"{tag}_dims"]]
    T, lr0, E = int(g["num_train_timesteps"]), float(g["lr"]), int(g["num_epochs"])
    sd = synth.synth_state_dict(synth.unet_spec(), int(g["unet_seed"]))
    tr = engine.UnetTrainer(sd, L)
    x0 = torch.from_numpy(g[f"{tag}_x0"])
    for step in range(3):
        lr = lr0 if step < 2 else lr0 * (1 + math.cos(math.pi / E)) / 2         # CosineAnnealingLR(T_max=num_epochs, eta_min=0) after one epoch
        assert abs(lr - float(g[f"{tag}_lr{step}"])) < 1e-12
        noise, ts = torch.from_numpy(g[f"{tag}_noise{step}"]), torch.from_numpy(g[f"{tag}_t{step}"])
        sa, sb = _coefs(T, ts)
        loss, _ = tr.step(x0, noise, sa, sb, ts, lr)
        assert abs(loss - float(g[f"{tag}_loss{step}"])) < 2e-5 * abs(float(g[f"{tag}_loss{step}"])), (step, loss, float(g[f"{tag}_loss{step}"]))
        if step == 0:
            grads = tr.export(1)
            worst = 0.0
            for k, v in grads.items():
                f = v.double().flatten().numpy()
                ref, (s1, s2) = g[f"{tag}_grad/{k}"], g[f"{tag}_gradsum/{k}"]
                rms = math.sqrt(s2 / f.size)
                err = np.abs(f[sample_idx(k, f.size)] - ref).max() / rms
                worst = max(worst, err)
                assert err < 5e-5, (k, err)
                assert abs((f * f).sum() - s2) < 1e-4 * s2, (k, (f * f).sum(), s2)
            print(f"{tag}: worst sampled gradient entry error / tensor rms {worst:.2e}")
    final = tr.export(0)
    flips, worst = 0, 0.0
    for k, v in final.items():
        f = v.double().flatten().numpy()
        d = np.abs(f[sample_idx(k, f.size)] - g[f"{tag}_final/{k}"])
        flips += int((d > 4e-6).sum())
        assert d.max() < 3 * 2 * lr0 + 1e-6, (k, d.max())
        worst = max(worst, float(d[d <= 4e-6].max()) if (d <= 4e-6).any() else 0.0)
        s1, s2 = g[f"{tag}_finalsum/{k}"]
        assert abs((f * f).sum() - s2) < 1e-5 * s2 + 1e-12, (k, (f * f).sum(), s2)
    assert flips <= 2, flips
    print(f"{tag}: parameters after 3 steps within {worst:.2e} of the reference's ({flips} sign-of-rounding entries)")


@pytest.mark.parametrize("B,L", [(3, 14), (130, 14), (129, 42)])
def test_unet_trainer_vs_oracle(B, L):
    """Ragged batch sizes (one sample tile and a half, odd counts): loss, noise prediction and EVERY gradient tensor against the oracle's
    autograd on the same draws; then one Adam + EMA step."""
    from dgdm_amd import engine, synth
    from oracle import dgdm_oracle as orc
    sd = synth.synth_state_dict(synth.unet_spec(), 5)
    o = orc.UnetTrainer(sd, 15, L, 1e-4, ema_power=0.85)
    tr = engine.UnetTrainer(sd, L)
    rs = np.random.RandomState(B * 100 + L)
    x0 = torch.from_numpy(rs.uniform(-1, 1, (B, L, 1)).astype(np.float32))
    torch.manual_seed(B + L)
    for step in range(2):
        torch.set_num_threads(min(8, os.cpu_count() or 1))
        lo, po = o.step(x0)
        noise, ts = o.draws
        sa, sb = _coefs(15, ts)
        lh, ph = tr.step(x0, noise, sa, sb, ts, 1e-4, want_pred=True)
        assert abs(lh - lo) < 2e-5 * abs(lo), (lh, lo)
        assert _rel(ph.cpu(), po) < 2e-5
        gh = tr.export(1)
        worst = max(_rel(gh[k], o.grads[k]) for k in gh)
        assert worst < 2e-4, sorted(((_rel(gh[k], o.grads[k]), k) for k in gh), reverse=True)[:5]
        tr.ema_step(o.ema.decay)       # the oracle stepped its EMA with this decay inside step()
        print(f"B={B} L={L} step {step}: loss {lh:.6f} (oracle {lo:.6f}), worst gradient tensor rel L2 {worst:.2e}")
    ph, eh = tr.export(0), tr.export(4)
    for k in ph:
        assert float((ph[k] - o.sd[k]).abs().max()) < 2.5e-4, k        # <= 2 lr (+ rounding) wherever a sign is decided by rounding
        assert float((eh[k] - o.ema.averaged[k]).abs().max()) < 2.5e-4, k
    frac = sum(int(((ph[k] - o.sd[k]).abs() > 4e-6).sum()) for k in ph) / sum(v.numel() for v in ph.values())
    assert frac < 1e-4, frac
    assert tr.steps() == 2


def test_unet_trainer_deterministic_and_resume():
    """Same inputs, same bits; export -> import (parameters, moments, step count) continues identically."""
    from dgdm_amd import engine, synth
    sd = synth.synth_state_dict(synth.unet_spec(), 9)
    B, L = 37, 14
    rs = np.random.RandomState(3)
    x0 = torch.from_numpy(rs.uniform(-1, 1, (B, L, 1)).astype(np.float32))
    noise = torch.from_numpy(rs.normal(size=(B, L, 1)).astype(np.float32))
    ts = torch.from_numpy(rs.randint(0, 15, B))
    sa, sb = _coefs(15, ts)
    a, b = engine.UnetTrainer(sd, L), engine.UnetTrainer(sd, L)
    for _ in range(2):
        la, _ = a.step(x0, noise, sa, sb, ts, 1e-4)
        lb, _ = b.step(x0, noise, sa, sb, ts, 1e-4)
        assert la == lb
    pa, pb = a.export(0), b.export(0)
    assert all(torch.equal(pa[k], pb[k]) for k in pa)
    c = engine.UnetTrainer(pa, L)
    c.load(2, a.export(2))
    c.load(3, a.export(3), adam_steps=a.steps())
    la, _ = a.step(x0, noise, sa, sb, ts, 5e-5)
    lc, _ = c.step(x0, noise, sa, sb, ts, 5e-5)
    assert la == lc
    pa, pc = a.export(0), c.export(0)
    assert all(torch.equal(pa[k], pc[k]) for k in pa)
