"""north_star tolerance at the shipped per-finger grids: free-running guided chains of the HIP path against the REFERENCE's own
``Diffusion.guided_sample`` / ``guided_sample_multi_object`` (tests/golden/g9_*.npz, made by tests/golden/make_golden.py g9_2d / g9_3d).

Per-finger statistics depend on the cell count C = G*P^2, not on B, so the fixtures use the BASELINE grids at a small batch:
  2-D  B=4, G=360, P=5 (C = 9000 cells per finger as in configs[1]; R = 36 000 rows per cond_fn), 100-vertex object
  3-D  B=2, G=45,  P=5 (C = 1125 cells per finger as in configs[2]; R = 2250 rows = 5 sub-batches of sub_bs = 512), 512-point object

Every fixture chain was run twice by the reference (8 CPU threads and 1 (2-D) / 4 (3-D) threads, same seeds and FPS draws); the
distance between those two end points is stored as ``floor``: the reference's own reproducibility.  With the synthetic He-init
dynamics weights as they are (``gain`` 1, the '*_raw' chains) the guidance term is 10^2..10^4 times eps, the chain is chaotic and
the reference parts from itself by 0.2 .. 4.5 - no implementation can be compared free-running there, only step by step.  The
other chains scale the dynamics output layer (``dgdm_amd.synth.scale_output``) so that the guidance term is of eps' order, as
the reference's classifier scales assume; there the reference reproduces itself to 1e-5 .. 7e-5 and the HIP path must be within

    finger L2  <  1e-4                       (north_star)      when floor <  3e-5
    finger L2  <  max(1e-4, 3 * floor)                          when floor >= 3e-5 (the reference itself is not better than that)

Every chain - chaotic or not - is also replayed step by step on the reference's recorded trajectory (eps-net, cond_fn gradient,
scheduler step: relative 2e-5), which is the precise check.  The measured numbers are printed (pytest -s) and tabulated in DESIGN.md.
"""
import json
import os

import numpy as np
import pytest
import torch

from dgdm_amd import engine, sampler, synth
from tests import util
from tests.test_gpu_parity import _teacher_forced, dev, finger_l2, sched      # noqa: F401  (dev is a fixture)

pytestmark = pytest.mark.gpu
NORTH_STAR = 1e-4
REPORT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "fullgrid_parity.json")


def tolerance(floor):
    """See the module docstring; None = the reference does not reproduce itself (chaotic chain): no free-running comparison."""
    if floor > 5e-2:
        return None
    return NORTH_STAR if floor < 3e-5 else max(NORTH_STAR, 3.0 * floor)


def _report(rows):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        old = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        old.update(rows)
        json.dump(old, open(REPORT, "w"), indent=1)
    except OSError:
        pass


def test_fullgrid_2d(dev):
    g = util.load("g9_2d.npz")
    B, G, P, L, T, S, nv = [int(v) for v in g["dims"]]
    assert (G, P) == (360, 5) and nv == 100
    net = engine.Unet1d(util.unet_sd(g["unet_seed"]))
    objs = torch.from_numpy(g["objs"])
    s = sched(T, S)
    noise = synth.synth_noise(0, B, L).to(dev)
    ug = sampler.unguided_sample(net, s, noise)
    assert finger_l2(ug.cpu(), g["unguided"]) < NORTH_STAR
    rows = {}
    for name in [str(n) for n in g["names"]]:
        o, gain, floor = str(g[f"{name}_opt_obj"]), float(g[f"{name}_gain"]), float(g[f"{name}_floor"])
        dyn = engine.Dynamics(2, synth.scale_output(util.dyn2d_sd(g["dyn2d_seed"], nv), gain), L, 2 * nv)
        gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, nv, 0, max_objects=2)
        gd.set_objects(objs.to(dev))
        ref = g[f"{name}_guided"]
        errs = []
        if name == "multi":
            out = sampler.guided_multi_object(net, gd, s, 'point', noise, [0, 1], o).cpu()
            _teacher_forced(net, gd, s, 'point', g, f"{name}_trace", [(0, o)], ref, None, dev, scale=sampler.SCALE_2D, multi_obj=[0, 1], errs=errs)
        else:
            out = sampler.guided_chains(net, gd, s, 'point', noise, [(0, o)], unguided=ug)[0].cpu()
            rc = None
            if o == 'convergence':
                centers = sampler.convergence_centers(gd, 'point', ug, [0])
                assert np.array_equal(centers[0].numpy(), g[f"{name}_centers"])
                rc = torch.from_numpy(gd.rowcoef(centers[0])).to(dev).reshape(1, -1)
            _teacher_forced(net, gd, s, 'point', g, f"{name}_trace", [(0, o)], ref, None, dev, scale=sampler.classifier_scale('point', o), rowcoef=rc, errs=errs)
        err, tol = finger_l2(out, ref), tolerance(floor)
        rows[f"2d/{name}"] = dict(opt_obj=o, gain=gain, reference_floor=floor, hip_vs_reference=err, tolerance=tol, max_step_grad_rel=max(errs))
        print(f"2d {name:16s} gain {gain:.4g} reference floor {floor:.2e} HIP vs reference {err:.2e} tol {tol} max per-step grad rel {max(errs):.1e}")
        if tol is not None:
            assert err < tol, (name, err, tol, floor)
    _report(rows)
    # the north-star bound proper must hold on the well-conditioned chains
    assert all(r["hip_vs_reference"] < NORTH_STAR for k, r in rows.items() if r["reference_floor"] < 3e-5)
    assert sum(r["reference_floor"] < 3e-5 for r in rows.values()) >= 3


def _load3d(part):
    f = os.path.join(util.GOLDEN, f"g9_3d_{part}.npz")
    return np.load(f) if os.path.exists(f) else None


@pytest.mark.parametrize("part", ["rotate", "convergence", "multi", "rotate_raw"])
def test_fullgrid_3d(dev, part):
    g = _load3d(part)
    if g is None:
        pytest.skip(f"tests/golden/g9_3d_{part}.npz has not been generated")
    alt = _load3d(part + "_alt")
    floor = float(alt["floor"]) if alt is not None else None
    B, G, P, L, T, S, N = [int(v) for v in g["dims"]]
    assert (G, P, N) == (45, 5, 512)
    o, gain = str(g["opt_obj"]), float(g["gain"])
    net = engine.Unet1d(util.unet_sd(g["unet_seed"]))
    dyn = engine.Dynamics(3, synth.scale_output(util.dyn3d_sd(g["dyn3d_seed"]), gain), L)
    objs = torch.from_numpy(g["objs"])
    gd = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), 2, T, N, 512, max_objects=2)
    gd.set_objects(objs.to(dev))
    s = sched(T, S)
    noise = synth.synth_noise(0, B, L).to(dev)
    ug = sampler.unguided_sample(net, s, noise)
    assert finger_l2(ug.cpu(), g["unguided"]) < NORTH_STAR
    forced = lambda: sampler.StartStream(N, 512, util.unpack_starts(g["starts"].astype(np.int64), g["start_lens"]))      # noqa: E731
    ref = g["guided"]
    errs = []
    if part == "multi":
        out = sampler.guided_multi_object(net, gd, s, 'point_3d', noise, [0, 1], o, starts=forced()).cpu()
        st = forced()
        step = np.stack([np.stack([st.call(gd.rows), st.call(gd.rows)]) for _ in range(S)])
        _teacher_forced(net, gd, s, 'point_3d', g, "trace", [(0, o)], ref, step, dev, scale=sampler.SCALE_3D, multi_obj=[0, 1], errs=errs)
    else:
        out = sampler.guided_chains(net, gd, s, 'point_3d', noise, [(0, o)], unguided=ug, starts=forced())[0].cpu()
        sweep, step = sampler.draw_chain_starts(gd, [(0, o)], S, forced())
        rc = None
        if o == 'convergence':
            centers = sampler.convergence_centers(gd, 'point_3d', ug, [0], sweep[0])
            rc = torch.from_numpy(gd.rowcoef(centers[0])).to(dev).reshape(1, -1)
        _teacher_forced(net, gd, s, 'point_3d', g, "trace", [(0, o)], ref, step, dev, scale=sampler.classifier_scale('point_3d', o), rowcoef=rc, errs=errs)
    err = finger_l2(out, ref)
    tol = tolerance(floor) if floor is not None else NORTH_STAR
    _report({f"3d/{part}": dict(opt_obj=o, gain=gain, reference_floor=floor, hip_vs_reference=err, tolerance=tol, max_step_grad_rel=max(errs))})
    print(f"3d {part:12s} gain {gain:.4g} reference floor {floor} HIP vs reference {err:.2e} tol {tol} max per-step grad rel {max(errs):.1e}")
    if tol is not None:
        assert err < tol, (part, err, tol, floor)
