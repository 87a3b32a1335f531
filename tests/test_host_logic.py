"""CPU tests of the host-side logic of the product path (no GPU, no HIP compute calls)."""
import ctypes as C
import os
import re
import shlex

import numpy as np
import pytest
import torch

from dgdm_amd import _lib, sampler, synth
from oracle import dgdm_oracle as orc
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dgdm_amd import build
    build.build()
    return _lib.lib()


def test_abi_exports_every_declared_symbol(lib):
    """include/dgdm_hip.h <-> libdgdm_hip.so <-> the ctypes prototype table agree on the symbol set."""
    hdr = open(os.path.join(ROOT, "include", "dgdm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dgdm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in dgdm_hip.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.dgdm_version() == 100
    assert C.sizeof(_lib.Objective) == 32 and C.sizeof(_lib.Tensor) == 32 and C.sizeof(_lib.GuidanceConfig) == 40


def test_objective_table_matches_oracle(lib):
    d = torch.tensor([[0.3, -0.7, 1.1]], requires_grad=True)
    names = list(orc._LINEAR_OBJ) + ['rotate']
    for n in names:
        o = _lib.Objective()
        _lib.check(lib.dgdm_objective_from_name(n.encode(), C.byref(o)))
        g = torch.autograd.grad(orc.deltas_to_objective(d, n).sum(), d)[0][0]
        mine = torch.tensor([o.lin[j] + 2 * o.quad[j] * float(d[0, j]) for j in range(3)])
        assert torch.allclose(g, mine), n
    o = _lib.Objective()
    with pytest.raises(ValueError, match='opt obj not supported'):
        _lib.check(lib.dgdm_objective_from_name(b"spin", C.byref(o)))


@pytest.mark.parametrize("rows,sub", [(24 * 4 * 3, 0), (150, 64), (150, 7), (1125 * 2, 512)])
def test_convergence_rowcoef_matches_autograd(lib, rows, sub):
    """dgdm_convergence_rowcoef == d objective / d delta_0 of the oracle's 'convergence' objective, per cond_fn call."""
    G, P = 6, 2
    cells = G * P * P
    rs = np.random.RandomState(rows + sub)
    centers = rs.randint(0, G, size=rs.randint(1, 9)).astype(np.int64)
    out = np.empty(rows, np.float32)
    _lib.check(lib.dgdm_convergence_rowcoef(centers.ctypes.data, len(centers), G, P, rows, sub, out.ctypes.data))
    ref = np.zeros(rows, np.float32)
    step = sub if sub > 0 else rows
    for s0 in range(0, rows, step):
        n = min(step, rows - s0)
        d = torch.zeros(n, 3, requires_grad=True)
        obj = orc.deltas_to_objective(d, 'convergence', centers=torch.from_numpy(centers), grid_size=G, num_pos=P)
        if obj.numel():
            ref[s0:s0 + n] = torch.autograd.grad(obj.sum(), d)[0][:, 0].numpy()
    assert np.array_equal(out, ref)
    assert cells > 0


def test_start_stream_matches_reference_draws():
    """StartStream replays torch.randint exactly as the oracle's (= the reference's) classifier calls consume it."""
    g = util.load("g5_dyn3d.npz")
    for sub, rows in ((7, 24), (512, 24)):
        torch.manual_seed(99)
        got = sampler.StartStream(512, sub).call(rows)
        assert np.array_equal(got, g[f"starts_rotate_sub{sub}"])
    torch.manual_seed(5)
    a = sampler.StartStream(300, 64).call(150)          # N != 512: per-call draws with the two ranges
    torch.manual_seed(5)
    log = orc.StartLog()
    for r0 in range(0, 150, 64):
        n = min(64, 150 - r0)
        log.draw(300, n)
        log.draw(512, n)
    assert np.array_equal(a, torch.cat(log.log).numpy())


def test_torch_rng_replay_is_bit_exact(lib):
    """The library's replay of torch's CPU generator (csrc/torch_rng.hip) against torch itself: seeding, randint with power-of-two
    and other ranges across state refreshes, skipping, and the state blob it hands back."""
    for seed in (0, 1234, 2 ** 40 + 7):
        g = torch.Generator()
        g.manual_seed(seed)
        r = sampler.TorchRng(seed=seed)
        for high, n in ((512, 5), (512, 700), (300, 1000), (1000, 3), (512, 100000), (7, 1300), (512, 0)):
            assert np.array_equal(torch.randint(0, high, (n,), generator=g).numpy(), r.randint(high, n)), (seed, high, n)
        torch.randint(0, 512, (12345,), generator=g)
        assert r.randint(512, 12345, skip=True) is None
        assert np.array_equal(torch.randint(0, 99, (50,), generator=g).numpy(), r.randint(99, 50))
        assert np.array_equal(g.get_state().numpy(), r._blob)                      # torch could take over from here
    # attached to the global generator: torch and the replay alternate on one stream
    torch.manual_seed(42)
    a = [torch.randint(0, 512, (n,)).numpy() for n in (100, 1000, 10)]
    torch.manual_seed(42)
    r = sampler.TorchRng()
    b = [r.randint(512, 100), torch.randint(0, 512, (1000,)).numpy(), r.randint(512, 10)]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # a private torch.Generator
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    st = sampler.StartStream(300, 64, generator=g2)
    want = []
    for r0 in range(0, 150, 64):
        n = min(64, 150 - r0)
        want += [torch.randint(0, 300, (n,), generator=g1), torch.randint(0, 512, (n,), generator=g1)]
    assert np.array_equal(st.call(150), torch.cat(want).numpy())
    assert torch.equal(g1.get_state(), g2.get_state())
    # strided destination: one chain's column of a [step][chain][2 * rows] array
    out = np.full((3, 4, 2 * 40), -1, dtype=np.int64)
    sampler.StartStream(512, 16, seed=5).calls(40, 3, out=out[:, 2])
    assert np.array_equal(out[:, 2], sampler.StartStream(512, 16, seed=5).calls(40, 3)) and (out[:, [0, 1, 3]] == -1).all()


def test_start_plan_draws_ahead_like_torch():
    """sampler.StartPlan: the draws of a list of jobs made ahead on a worker thread == the same StartStream calls made one after the
    other on torch's global generator, kept or skipped; the generator ends where torch.randint would have left it; a request that is
    not the planned one cancels the plan and continues synchronously from the right place."""
    jobs = [(24, 3, True), (10, 1, False), (24, 5, True), (7, 2, True)]

    def run(plan):
        torch.manual_seed(77)
        st = sampler.StartStream(512, 7)
        got = []
        if plan:
            with sampler.StartPlan(512, 7, jobs):
                got = [st.calls(r, n, keep=k) for r, n, k in jobs]
        else:
            got = [st.calls(r, n, keep=k) for r, n, k in jobs]
        return got, torch.randint(0, 1000, (5,))

    (a, ta), (b, tb) = run(False), run(True)
    assert all((x is None and y is None) or np.array_equal(x, y) for x, y in zip(a, b)) and torch.equal(ta, tb)
    # against torch.randint itself (N = 512: a call is one flat run of 2 * rows draws)
    torch.manual_seed(77)
    for (r, n, k), got in zip(jobs, b):
        want = torch.randint(0, 512, (2 * r * n,)).numpy().reshape(n, 2 * r)
        assert got is None if not k else np.array_equal(got, want)
    # deviation from the plan
    torch.manual_seed(3)
    ref = [sampler.StartStream(512, 7).calls(24, 3), sampler.StartStream(512, 7).calls(5, 1), torch.randint(0, 9, (4,))]
    torch.manual_seed(3)
    with sampler.StartPlan(512, 7, jobs):
        st = sampler.StartStream(512, 7)
        x = st.calls(24, 3)
        y = st.calls(5, 1)                   # not the planned (10, 1, skip): the plan is dropped, the stream goes on from after job 0
    z = torch.randint(0, 9, (4,))
    assert np.array_equal(x, ref[0]) and np.array_equal(y, ref[1]) and torch.equal(z, ref[2])
    assert sampler._ACTIVE_PLAN is None


def test_scheduler_tables_match_oracle():
    from dgdm_amd.scheduler import DDIMScheduler
    for T, S in ((15, 5), (1000, 100), (1000, 1000)):
        s, o = DDIMScheduler(num_train_timesteps=T), orc.DDIM(T)
        s.set_timesteps(S)
        o.set_timesteps(S)
        assert torch.equal(s.betas, o.betas) and torch.equal(s.alphas_cumprod, o.alphas_cumprod) and torch.equal(s.timesteps, o.timesteps)
        assert s.config.num_train_timesteps == T
        for t in (int(s.timesteps[0]), int(s.timesteps[-1])):
            prev = t - T // S
            a_t, a_p = o.alphas_cumprod[t], (o.alphas_cumprod[prev] if prev >= 0 else o.final_alpha_cumprod)
            want = (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5))
            assert s.coefficients(t) == want


def test_metrics_match_golden():
    from dgdm_amd.dynamics import metrics
    g = util.load("g7_convergence.npz")
    for k in ("all0", "all2", "all1", "wrap", "mixed", "single", "ones_between"):
        l, c = metrics.convergence_mode_three_class(torch.from_numpy(g[f"{k}_profile"]))
        assert np.array_equal(l.numpy(), g[f"{k}_lengths"]) and np.array_equal(c.numpy(), g[f"{k}_centers"]), k
    a = torch.arange(10.0)
    for i in range(6):
        lo, hi = g[f"slicer_{i}_args"]
        assert np.array_equal(metrics.slicer(a, int(lo), int(hi)).numpy(), g[f"slicer_{i}"])


SHIPPED_3D = ("--mode='test' --checkpoint_path='ckpts/dynamics_3d.pt' --diffusion_checkpoint_path='ckpts/diffusion_3d.ckpt' --object_dir='' "
              "--save_dir='' --classifier_guidance --num_fingers=16 --grid_size=45 --num_pos=5 --fingers_3d --object_max_num_vertices=512 "
              "--ctrlpts_dim=42 --ctrlpts_x_dim=7 --ctrlpts_z_dim=3 --num_workers=0 --num_train_timesteps=15 --num_inference_steps=5 "
              "--ema_power=0.85 --batch_size=16 --sub_bs=512 --num_cpus=32 --seed=0")
SHIPPED_2D = ("--mode='test' --checkpoint_path='ckpts/dynamics_2d.pt' --classifier_guidance --diffusion_checkpoint_path='ckpts/diffusion_2d.pt' "
              "--object_dir='x/Icons-50.npy' --save_dir='' --ctrlpts_dim=14 --num_fingers=16 --grid_size=360 --num_pos=5 "
              "--object_max_num_vertices=100 --num_workers=0 --num_train_timesteps=15 --num_inference_steps=5 --ema_power=0.85 "
              "--batch_size=16  --num_cpus=32 --seed=0")


def test_parser_accepts_the_shipped_command_lines():
    """The flag set of generator/guided_sample_{2d,3d}.sh parses; defaults equal the reference's (dynamics/parser.py)."""
    from dynamics.parser import parse       # the reference's import path, served by the repo-root shim
    a = parse(shlex.split(SHIPPED_3D))
    assert (a.fingers_3d, a.grid_size, a.num_pos, a.sub_bs, a.ctrlpts_dim, a.batch_size) == (True, 45, 5, 512, 42, 16)
    b = parse(shlex.split(SHIPPED_2D))
    assert (b.fingers_3d, b.grid_size, b.object_max_num_vertices, b.num_train_timesteps) == (False, 360, 100, 15)
    d = parse([])
    assert (d.batch_size, d.sub_bs, d.num_pos, d.num_inference_steps, d.ema_power, d.seed, d.save_dir) == (1024, 1024, 9, 100, 0.75, 0, None)


def test_module_trees_carry_reference_state_dict_keys():
    from generator.diffusion_utils import ConditionalUnet1D
    from dynamics.profile_forward_2d import ProfileForward2DModel
    from dynamics.profile_forward_3d import ProfileForward3DModel
    for m, spec in ((ConditionalUnet1D(1, 0, [128, 256], 32), synth.unet_spec()),
                    (ProfileForward2DModel(params_ch=14, object_ch=200), synth.dyn2d_spec(14, 200)),
                    (ProfileForward3DModel(params_ch=42), synth.dyn3d_spec(42))):
        sd = m.state_dict()
        assert sorted(sd) == sorted(k for k, _ in spec)
        assert all(tuple(sd[k].shape) == tuple(s) for k, s in spec)
    # and there is no CPU path: asking for a forward without a GPU fails loudly
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            ConditionalUnet1D(1, 0, [128, 256], 32)(torch.zeros(1, 14, 1), torch.zeros(1, dtype=torch.int64))


def test_finger_dataset_is_the_reference_recipe():
    from dgdm_amd.generator.train import finger_control_points
    from generator.dataloader import GripperDataset
    pts = finger_control_points(4, True)
    rs = np.random.RandomState(2)
    yl, yr = rs.uniform(-0.1, 0, size=21), rs.uniform(-0.1, 0, size=21)
    assert np.array_equal(pts[2, :21, 1], yl) and np.array_equal(pts[2, 21:, 1], yr) and pts.shape == (4, 42, 3)
    item = GripperDataset(pts, 0.12, -0.12, 0.0, -0.1)[2]
    assert item.shape == (42, 1) and item.dtype == np.float32 and np.abs(item).max() <= 1.0
    assert np.allclose(item[:21, 0], (yl.astype(np.float32) + 0.1) / 0.1 * 2 - 1)


def test_trainer_draws_ahead_and_adopts_them():
    """Trainer._draw (dynamics/trainer.py:68-74 of the reference: torch.randn for the noise, then torch.randint for the timesteps, on the
    global CPU generator): after every call a worker draws the next call's numbers; the next call must ADOPT them (not draw again) when
    nobody touched the generator in between, throw them away when somebody did or the row count changed, and either way hand back exactly
    what the reference's two calls give."""
    import argparse
    from dgdm_amd.dynamics.trainer import Trainer
    t = Trainer(argparse.Namespace(use_sub_batch=False, sub_bs=1024, grid_size=360, learning_rate=1e-4, weight_decay=0.0, num_epochs=100,
                                   checkpoint_path=None, fingers_3d=False, ctrlpts_dim=14, object_max_num_vertices=100,
                                   num_timesteps_per_batch=1, num_inference_steps=5, num_train_timesteps=15))
    rows = 1000

    def ref(n):
        return torch.randn((n, 14)), torch.randint(0, 15, (n,)).long()

    torch.manual_seed(5)
    want = [ref(rows), ref(rows), torch.rand(3), ref(rows), ref(70), ref(70)]
    torch.manual_seed(5)
    def draw(n, adopted):
        pending = t._ahead[1] if t._ahead is not None else None
        noise, ts = t._draw(n)
        assert (pending is not None and noise.data_ptr() == pending["noise"].data_ptr()) == adopted       # the very buffer the worker filled
        return noise.clone(), ts      # (the worker's two pinned buffers alternate: the caller uploads a draw before the one after next is made)

    got = [draw(rows, False)]
    assert t._ahead is not None                                   # the worker is on its way
    got.append(draw(rows, True))
    got.append(torch.rand(3))                                     # somebody else draws: the pending numbers are stale
    got.append(draw(rows, False))
    got.append(draw(70, False))                                   # another row count: stale as well
    got.append(draw(70, True))
    t._join_ahead()
    for w, g in zip(want, got):
        if isinstance(w, tuple):
            assert torch.equal(w[0], g[0]) and torch.equal(w[1], g[1])
        else:
            assert torch.equal(w, g)


def test_bench_headline_is_compact_and_parses():
    """The bench contract's one JSON line, built from canned numbers: under 4 KB (round 4's 20 KB line was cut by the driver's tail
    capture and parsed as nothing), scalars only inside roofline / cpu_baseline, and it carries every key the driver reads."""
    import json
    import bench
    m = {"samples": 20480, "secs": 1.2345678901, "world": 8, "steps": 20, "warmup": 5, "contraction": "f32", "workload": "3d", "pairs": 32, "B": 32, "S": 5,
         "rows": 36000, "n_obj": 1, "draw_secs": 0.0123456789, "gloo": True}
    roof = {"bound": "mfma", "kernel": "trunk_f16l_kernel", "achieved": 376.61438352316435, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.15064575340926575,
            "traffic": 1742654688.0, "traffic_fetch_bytes_raw": 1667157216.0, "traffic_write_bytes": 75497472.0, "traffic_vs_algorithmic": 1.4683048612796248,
            "traffic_recorded": "profiles/r04_3d_pmc_hbm.json", "launches": 5, "avg_launch_ms": 7.2166893005371096, "algorithmic_flops_per_launch": 2717908992000.0,
            "issued_flops_per_algorithmic_flop": 3, "matrix_pipe_issue_frac": 0.4519372602277972, "frac_algorithmic_vs_f32_peak": 2.394242743313187,
            "pipe_busy_recorded": 0.54, "arithmetic": "f32_f16x3", "share_of_step": 0.5735658003847591, "step_frac": 0.09070971299343188,
            "step_issue_frac": 0.263520217250933, "step_necessary_tflop": 14.26653696}
    stages = {"trunk": 36.0834, "unet": 7.0088, "xobj": 7.4601, "tables": 11.5487, "guide_misc": 1.4827, "ddim": 0.0324, "profiled_step_wall": 64.45}
    cpu = {"value": 0.0064026135, "unit": "samples/s", "cores": 32, "kind": "port", "sample": "1 run of 1 of the 71 sub-batches (512 of 36000 replicated rows) of one "
           "cond_fn call (PointNet++ + trunk fwd, autograd bwd) + 1 eps-net forward, extrapolated to 36000 rows x 5 steps", "ms_per_denoise_step": 999591.8,
           "cpu_seconds": 14.2}
    line = json.dumps(bench.headline(m, roof, stages, cpu))
    assert len(line) < 4096 and "\\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 8 and d["vs_baseline"] is None and d["scaling"] == "weak" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 20480 / 1.2345678901) < 1e-2 and abs(d["ms_per_step"] - 1234.5678901 / 20) < 1e-4
    assert d["roofline"]["frac"] == pytest.approx(0.15064575, rel=1e-6) and d["roofline"]["bound"] == "mfma"
    assert all(not isinstance(v, (dict, list)) for v in d["roofline"].values()) and all(not isinstance(v, (dict, list)) for v in d["cpu_baseline"].values())
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    # every workload / contraction label stays short enough
    for w in bench.WORKLOAD_NAME:
        for c in bench.DTYPE_LABEL:
            assert len(json.dumps(bench.headline(dict(m, workload=w, contraction=c), roof, stages, cpu))) < 4096


def test_shipped_library_is_built_from_this_tree():
    """Build provenance: dgdm_amd/csrc/libdgdm_hip.so is git-ignored and travels to the GPU box with the snapshot; its side file records the
    content hashes of every source / header / flag set it was compiled from, and rebuilds are decided by those hashes (not by mtimes).  The
    library that the tests load is the one built from the sources in this tree."""
    import hashlib
    from dgdm_amd import build as b
    info = b.build_info()
    assert info, "no library (or no provenance file) - run `python __graft_entry__.py`"
    assert info["current"], "libdgdm_hip.so is older than the sources next to it"
    assert hashlib.sha256(hashlib.sha256(open(b.LIB, "rb").read()).digest()).hexdigest() == info["lib_sha256"]
    assert set(info["sources"]) == set(b.SOURCES)
