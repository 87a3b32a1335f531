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


def test_diffusion_training_api():
    """The reference-shaped calls (generator/diffusion.py:126-177, 711-724) on the mirror class: configure_optimizers, get_stats (no update),
    training_step + on_train_batch_end (update + EMA), against the oracle fed the SAME device-generator draws; then the trained
    weights are what validation_step's eps-net evaluates, and a checkpoint round trip continues bit for bit."""
    from dgdm_amd import synth
    from dgdm_amd.generator.diffusion import Diffusion
    from dgdm_amd.generator.diffusion_utils import ConditionalUnet1D
    from dgdm_amd.scheduler import DDIMScheduler
    from oracle import dgdm_oracle as orc
    dev = torch.device("cuda:0")
    B, L, T = 12, 14, 15
    sd = synth.synth_state_dict(synth.unet_spec(), 21)

    def make():
        net = ConditionalUnet1D(input_dim=1, global_cond_dim=0, down_dims=[128, 256], diffusion_step_embed_dim=32)
        net.load_state_dict(sd)
        d = Diffusion(noise_pred_net=net, noise_scheduler=DDIMScheduler(num_train_timesteps=T), num_inference_steps=5, mode='point', input_dim=1,
                      num_points=L, learning_rate=1e-4, ema_power=0.85).to(dev)
        d.configure_optimizers()
        return d
    d = make()
    o = orc.UnetTrainer(sd, T, L, 1e-4, ema_power=0.85)
    x0 = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, (B, L, 1)).astype(np.float32))
    torch.manual_seed(11)
    st = d.get_stats(x0)
    assert set(st) == {"loss", "lr"} and st["lr"] == 1e-4
    for step in range(3):
        torch.manual_seed(100 + step)
        loss = float(d.training_step(x0, step))
        d.on_train_batch_end(None, x0, step)
        torch.manual_seed(100 + step)                      # the same draws, in get_stats' order, from the same (device) generator
        noise = torch.randn((B, L, 1), device=dev).cpu()
        ts = torch.randint(0, T, (B,), device=dev).long().cpu()
        lo, _ = o.step(x0, forced=(noise, ts))
        assert abs(loss - lo) < 2e-5 * abs(lo), (step, loss, lo)
        assert abs(d.ema.decay - o.ema.decay) < 1e-12
    d.lr_scheduler.step()
    assert abs(d.optimizer.param_groups[0]["lr"] - 1e-4 * (1 + math.cos(math.pi / 10000)) / 2) < 1e-15      # T_max = the constructor's num_epochs default
    ema = d.ema.averaged_model
    assert max(float((ema[k] - o.ema.averaged[k]).abs().max()) for k in ema) < 2.5e-4
    # sampling sees the trained weights
    d.eval()
    out = d.validation_step(x0, 1)
    with torch.no_grad():
        xs = torch.from_numpy(np.random.RandomState(0).randn(B, L, 1).astype(np.float32))
        ref_eps = orc.unet1d_forward(o.sd, xs, torch.full((B,), 3, dtype=torch.int64))
    got = d.noise_pred_net(xs.to(dev), torch.full((B,), 3, device=dev)).cpu()
    assert _rel(got, ref_eps) < 5e-4 and "val/denoise loss" in out["stats"]        # parameters differ by <= 2 lr where Adam's sign is rounding's
    # checkpoint: Lightning-shaped, resumes bit for bit
    ck = d.checkpoint(epoch=1, global_step=3)
    assert "ema_model" in ck["state_dict"] and all(k.startswith("noise_pred_net.") for k in ck["state_dict"]["ema_model"])
    assert all(k.startswith("ema_nets.noise_pred_net.") for k in ck["state_dict"] if k != "ema_model")
    assert len(ck["optimizer_states"][0]["state"]) == len(list(d.noise_pred_net.parameters()))
    d2 = make()
    d2.load_checkpoint(ck)
    torch.manual_seed(5)
    la = float(d.training_step(x0, 3))
    d.on_train_batch_end()
    torch.manual_seed(5)
    lb = float(d2.training_step(x0, 3))
    d2.on_train_batch_end()
    assert la == lb
    pa, pb = d._trainer().export(0), d2._trainer().export(0)
    assert all(torch.equal(pa[k], pb[k]) for k in pa)
    ea, eb = d.ema.averaged_model, d2.ema.averaged_model
    assert all(torch.equal(ea[k], eb[k]) for k in ea)


def test_train_cli_end_to_end(tmp_path):
    """`python generator/train.py <flags of train_diffusion_2d.sh>` (reduced finger count / epochs): the training loss falls, a
    Lightning-shaped checkpoint per epoch appears, `--diffusion_checkpoint_path=last.ckpt` resumes, and the checkpoint loads into the
    sampling entry point (`--mode=test`)."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    save = tmp_path / "out"
    base = [sys.executable, "generator/train.py", "--num_fingers=640", f"--save_dir={save}", "--learning_rate=1e-3", "--lr_warmup_steps=0", "--val_step=2",
            "--num_workers=0", "--num_train_timesteps=15", "--num_inference_steps=5", "--ema_power=0.85", "--batch_size=64", "--ctrlpts_dim=14"]
    r = subprocess.run(base + ["--num_epochs=6"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if "train/loss=" in l]
    losses = [float(l.split("train/loss=")[1].split(",")[0]) for l in lines]
    assert len(losses) == 6 and losses[-1] < 0.6 * losses[0], losses
    assert any("val/denoise loss" in l for l in lines)
    ck = torch.load(save / "checkpoints" / "last.ckpt", weights_only=False)
    assert ck["epoch"] == 6 and ck["global_step"] == 6 * 9 and "ema_model" in ck["state_dict"]
    assert (save / "checkpoints" / "epoch=0005.ckpt").exists() and (save / "val_vis" / "1_0.png").exists()
    r = subprocess.run(base + ["--num_epochs=7", f"--diffusion_checkpoint_path={save / 'checkpoints' / 'last.ckpt'}"], cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if "train/loss=" in l]
    assert len(lines) == 1 and "epoch=6" in lines[0] and float(lines[0].split("train/loss=")[1].split(",")[0]) < 1.2 * losses[-1], lines
    r = subprocess.run([sys.executable, "generator/train.py", "--mode=test", "--num_fingers=4", "--batch_size=4", "--ctrlpts_dim=14", "--num_train_timesteps=15",
                        "--num_inference_steps=5", f"--diffusion_checkpoint_path={save / 'checkpoints' / 'last.ckpt'}", f"--save_dir={tmp_path / 'smp'}"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "diffusion checkpoint" not in r.stderr          # no fallback to synthetic weights: the trained checkpoint was read


def test_train_cli_two_ranks_hold_identical_replicas(tmp_path):
    """`--mode=train` under a launcher (two ranks sharing this box's GPU, gloo for the collectives): rank 0's random eps-net is broadcast
    before the first step (what Lightning's DDP wrap does, generator/train.py:147-162), every rank trains on its slice of the epoch's
    permutation with averaged gradients, and fit() itself checks at the end that the replicas are bit-identical (it raises otherwise)."""
    import socket
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "generator/train.py", "--num_fingers=320", f"--save_dir={tmp_path / 'out'}", "--learning_rate=1e-3", "--lr_warmup_steps=0",
           "--val_step=100", "--num_workers=0", "--num_train_timesteps=15", "--num_inference_steps=5", "--ema_power=0.85", "--batch_size=32",
           "--ctrlpts_dim=14", "--num_epochs=3"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DGDM_DIST_BACKEND="gloo",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    assert "2 ranks hold identical eps-net parameters" in outs[0][0], outs[0][0][-2000:]
    losses = [float(l.split("train/loss=")[1].split(",")[0]) for l in outs[0][0].splitlines() if "train/loss=" in l]
    assert len(losses) == 3 and losses[-1] < losses[0], losses
    assert "train/loss=" not in outs[1][0]                      # only rank 0 reports
