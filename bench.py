#!/usr/bin/env python3
"""Benchmark of the guided-sampling hot path (BASELINE.json metric: guided samples/sec over the full DDIM chain,
plus ms per denoise step) on MI355X.

Workload (default, BASELINE config 3 shape; config 4 at N = 8): 3-D dynamics-guided sampling.  One *step* of this
benchmark is one batch of `--pairs` independent (object x objective) pairs per GPU, each a complete guided chain of
B = 32 fingers: PointNet++ object tables, then S = 5 x [eps-net, cond_fn over R = 32*45*25 = 36 000 replicated rows with
sub_bs = 512 FPS-start partition, guidance combine, DDIM step].  Every pair has its own synthetic 512-point object, so
nothing is shared between pairs; the tables are rebuilt inside the timed region for every pair.  The FPS start
indices (the torch.randint draws of pointnet2_utils.py:83) are the path's random input: they are drawn on the host
in the reference's order by a background thread one step ahead and handed over as host buffers, so the timed region
contains their host->device copy but not the Mersenne-Twister draw.
`--workload 2d` runs BASELINE config 2 (B = 64, G = 360, P = 5, R = 576 000 rows per pair, 100-vertex contours).

Contract: `python bench.py --gpus N --steps K --warmup W`; one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE from the
environment when N > 1); W untimed steps, exactly K timed steps between barrier + synchronize; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dgdm_amd import _lib, engine, sampler, synth          # noqa: E402
from dgdm_amd.scheduler import DDIMScheduler               # noqa: E402
from dgdm_amd.dist import gather_pairs                     # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: v_mfma_f32_32x32x16_bf16, dense (not the 2:1-sparsity figure)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", choices=["3d", "2d", "3d_ensemble"], default="3d")
    p.add_argument("--pairs", type=int, default=0, help="(object x objective) pairs per GPU per step (default 32 for 3d, 4 for 2d; "
                                                        "3d_ensemble: chains per GPU per step, default 8, each averaging 4 objects' gradients)")
    p.add_argument("--contraction", choices=["f32", "bf16"], default="f32",
                   help="arithmetic of the dynamics-trunk contractions: f32 (the parity path, default) or bf16 operands with f32 accumulation")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip the secondary workload summary")
    return p.parse_args()


class Workload:
    def __init__(self, kind, pairs, dev, rank, contraction="f32"):
        self.kind, self.dev, self.contraction = kind, dev, contraction
        self.n_obj = 4 if kind == "3d_ensemble" else 1    # dynamics-gradient evaluations per chain and denoise step
        self.ensemble = kind == "3d_ensemble"
        if kind == "3d_ensemble":
            kind = self.kind = "3d"
        if kind == "3d":
            self.mode, self.B, self.G, self.P, self.L, self.N, self.sub = 'point_3d', 32, 45, 5, 42, 512, 512
        else:
            self.mode, self.B, self.G, self.P, self.L, self.N, self.sub = 'point', 64, 360, 5, 14, 100, 0
        self.T, self.S, self.pairs = 15, 5, pairs
        self.unet_sd = synth.synth_state_dict(synth.unet_spec(), 11)
        if kind == "3d":
            self.dyn_sd = synth.synth_state_dict(synth.dyn3d_spec(self.L), 33)
            self.dyn = engine.Dynamics(3, self.dyn_sd, self.L)
        else:
            self.dyn_sd = synth.synth_state_dict(synth.dyn2d_spec(self.L, 2 * self.N), 22)
            self.dyn = engine.Dynamics(2, self.dyn_sd, self.L, 2 * self.N)
        self.net = engine.Unet1d(self.unet_sd, contraction_dtype=contraction)
        nch = pairs * self.n_obj                           # gradient chains (= distinct objects) per launch
        self.guid = engine.Guidance(self.dyn, self.B, self.G, self.P, (-1.0, 1.0), nch, self.T, self.N, self.sub, max_objects=nch, contraction_dtype=contraction)
        self.sched = DDIMScheduler(num_train_timesteps=self.T)
        self.sched.set_timesteps(self.S)
        self.noise = synth.synth_noise(0, self.B, self.L).to(dev)
        self.rank = rank
        objectives = [o for o in synth.OBJECTIVES_12 if o != 'convergence']     # 'convergence' needs the sim-side unguided pass
        self.chains = lambda step: [(i, objectives[(step * pairs + i) % len(objectives)]) for i in range(pairs)]
        self.rows = self.guid.rows

    def objects(self, step):
        """Synthetic objects of this step's pairs (distinct for every pair, rank and step), already on the device."""
        n = self.pairs * self.n_obj
        base = (self.rank * 100_000 + step) * n
        mk = synth.synth_object_3d if self.kind == "3d" else synth.synth_object_2d
        return torch.stack([mk(base + i, self.N) for i in range(n)]).to(self.dev)

    def draw(self, step):
        if self.kind != "3d":
            return None
        if self.ensemble:
            return sampler.draw_ensemble_starts(self.guid, self.pairs, self.n_obj, self.S)
        return sampler.draw_chain_starts(self.guid, self.chains(step), self.S)

    def run(self, step, objs, predrawn):
        self.guid.set_objects(objs)                      # 3-D: builds the PointNet++ tables of every pair (timed)
        if self.ensemble:                                # chain k averages the gradients of objects 4k..4k+3 (diffusion.py:637-647)
            groups = [list(range(self.n_obj * k, self.n_obj * (k + 1))) for k in range(self.pairs)]
            return sampler.guided_multi_object_groups(self.net, self.guid, self.sched, self.mode, self.noise, groups,
                                                      [o for _, o in self.chains(step)], predrawn=predrawn)
        return sampler.guided_chains(self.net, self.guid, self.sched, self.mode, self.noise, self.chains(step), predrawn=predrawn)


def timed_loop(wl, steps, warmup, dist):
    """Returns (seconds for `steps` steps, last output).  Inputs of step k+1 are prepared while step k runs."""
    total = warmup + steps
    prepared = {}

    def prepare(k):
        prepared[k] = (wl.objects(k), wl.draw(k))

    prepare(0)
    out = None
    t0 = None
    for k in range(total):
        if k == warmup:
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        th = None
        if k + 1 < total:
            th = threading.Thread(target=prepare, args=(k + 1,))
            th.start()
        objs, pre = prepared.pop(k)
        out = wl.run(k, objs, pre)
        if dist is not None and dist.get_world_size() > 1:
            out = gather_pairs(out, wl.pairs * dist.get_world_size())   # the path's only collective: final samples
        if th is not None:
            th.join()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def cpu_baseline(wl):
    """The CPU oracle (a restatement of the reference's as-written dataflow) on this box's host cores, bounded sample."""
    from oracle import dgdm_oracle as orc
    # torch CPU kernels on these small/medium tensors get slower beyond a few dozen threads (256 threads: >10x slower
    # than 32 on the MI355X host), so the baseline uses at most 32 - the count is reported in `cores`
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cores = torch.get_num_threads()
    so = orc.DDIM(wl.T)
    so.set_timesteps(wl.S)
    B, L = wl.B, wl.L
    x = wl.noise.cpu()
    ts = torch.full((B,), int(so.timesteps[0]), dtype=torch.int64)
    t0 = time.perf_counter()
    with torch.no_grad():
        orc.unet1d_forward(wl.unet_sd, x, ts)
    t_unet = time.perf_counter() - t0
    cells = wl.G * wl.P * wl.P
    if wl.kind == "3d":
        # one full 512-row sub-batch of cond_fn (fwd + autograd), as generator/diffusion.py:495-498 runs 71 of per step
        s = orc.Setup('point_3d', wl.unet_sd, wl.dyn_sd, so, L, wl.G, wl.P, wl.sub)
        obj = synth.synth_object_3d(0, wl.N)
        xr = x.clone().requires_grad_(True)
        ori, pos = orc._pose_grid(s, B, (-1.0, 1.0))
        n = wl.sub                                       # one full sub-batch of 512 rows (10-20 s on 32 threads)
        t0 = time.perf_counter()
        with torch.enable_grad():
            pts = orc._pts3d(s, xr).repeat(cells, 1, 1)[:n]
            logits = orc.dyn3d_forward(wl.dyn_sd, pts, ori[:n], pos[:n], ts.repeat(cells)[:n].float() / wl.T,
                                       obj.t().unsqueeze(0).expand(n, -1, -1), None)
            torch.autograd.grad(orc.deltas_to_objective(logits, 'rotate').sum(), xr)
        t_rows = (time.perf_counter() - t0) / n
        chain = wl.S * (B * cells * t_rows + t_unet)
        sample = (f"1 of the {(B * cells + n - 1) // n} sub-batches ({n} of {B * cells} replicated rows) of one cond_fn call (PointNet++ + trunk "
                  f"forward, autograd backward) + 1 eps-net forward, extrapolated to {B * cells} rows x {wl.S} steps")
    else:
        s = orc.Setup('point', wl.unet_sd, wl.dyn_sd, so, L, wl.G, wl.P)
        obj = synth.synth_object_2d(0, wl.N)
        Bs = 4                                          # 4 of 64 fingers against the full 9000-cell grid (36 000 rows)
        t0 = time.perf_counter()
        orc.cond_fn(s, x[:Bs], ts[:Bs], 'rotate', obj)
        t_c = (time.perf_counter() - t0) * (B / Bs)
        chain = wl.S * (t_c + t_unet)
        sample = f"cond_fn on {Bs} of {B} fingers x all {cells} cells (36000 rows) + 1 eps-net forward, extrapolated x{B // Bs} x{wl.S} steps"
    return {"value": B / chain, "unit": "guided samples/s", "cores": cores, "kind": "port", "sample": sample,
            "ms_per_denoise_step": chain / wl.S * 1e3}


def pmc_traffic(workload, kernel="trunk_kernel"):
    """HBM bytes per trunk launch from the committed rocprofv3 PMC passes of this same command (profiles/r01_*_pmc_hbm.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs, counter unit KB).  FETCH_SIZE is the raw counter: on gfx950 it can under-count
    wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM), so the true read traffic lies between 1x and 2x of `fetch_bytes_raw`."""
    f = os.path.join(ROOT, "profiles", f"r01_{workload}_pmc_hbm.json")
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    key = next((k for k in d["fetch"] if kernel + "<" in k), None)
    if key is None:
        return None
    fe, wr = d["fetch"][key]["avg_KB"] * 1024.0, d["write"][key]["avg_KB"] * 1024.0
    return {"bytes_per_launch": fe + wr, "fetch_bytes_raw": fe, "write_bytes": wr, "source": os.path.relpath(f, ROOT)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if a.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if world > 1:       # ranks share the host: keep torch's CPU pools (input synthesis, FPS start draws) from oversubscribing it
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // (2 * world)))
    _lib.device_init(local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(1234 + rank)
    pairs = a.pairs or {"3d": 32, "2d": 4, "3d_ensemble": 8}[a.workload]
    wl = Workload(a.workload, pairs, dev, rank, a.contraction)

    engine.prof_enable(False)
    secs, _ = timed_loop(wl, a.steps, a.warmup, dist)
    if dist is not None:
        tmax = torch.tensor([secs], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        secs = float(tmax.item())
    # second pass, rank 0 only: HIP events around every launch of the dominant kernel (the fused trunk)
    roof = None
    if rank == 0:
        engine.prof_enable(True)
        objs, pre = wl.objects(0), wl.draw(0)
        torch.cuda.synchronize()
        wl.run(0, objs, pre)
        torch.cuda.synchronize()
        n, ms, flops = engine.prof_read()
        engine.prof_enable(False)
        if n:
            ach = flops / (ms * 1e-3) / 1e12
            peak = BF16_MFMA_PEAK_TFLOPS if a.contraction == "bf16" else F32_MFMA_PEAK_TFLOPS
            kname = "trunk_bf16_kernel" if a.contraction == "bf16" else "trunk_kernel"
            roof = {"bound": "mfma", "kernel": kname + " (fused dynamics trunk fwd+bwd)", "achieved": ach, "peak": peak,
                    "unit": "TFLOP/s", "frac": ach / peak, "traffic": pmc_traffic(wl.kind, kname), "launches": n, "avg_launch_ms": ms / n,
                    "algorithmic_flops_per_launch": flops / n, "share_of_step": (ms * 1e-3) / (secs / a.steps)}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    samples = wl.B * pairs * world * a.steps
    line = {
        "metric": "guided samples/sec (full DDIM chain)", "value": samples / secs, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": secs / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.contraction, "data": "synthetic (random-init checkpoints, synthetic objects, seeded noise; SURVEY.md §8(d))",
        "config": {"workload": ("3-D guided sampling with a 4x guidance ensemble (BASELINE configs[4], guided_sample_multi_object semantics): per GPU "
                                "and step %d chains x B=32 fingers, each step averaging the dynamics gradients of 4 objects (4 cond_fn per chain-step), "
                                "G=45, P=5 -> R=36000 rows per cond_fn, sub_bs=512, 512-point objects, T=15/S=5"
                                if a.workload == "3d_ensemble" else
                                "3-D dynamics-guided sampling (BASELINE configs[2]; configs[3] at 8 GPUs): per GPU and step %d (object x objective) "
                                "pairs x B=32 fingers, G=45, P=5 -> R=36000 rows per cond_fn, sub_bs=512, 512-point objects, T=15/S=5"
                                if a.workload == "3d" else
                                "2-D dynamics-guided sampling (BASELINE configs[1]): per GPU and step %d (object x objective) pairs x B=64 fingers, "
                                "G=360, P=5 -> R=576000 rows per cond_fn, 100-vertex objects, T=15/S=5") % pairs,
                   "pairs_per_gpu_per_step": pairs, "fingers_per_pair": wl.B, "denoise_steps": wl.S, "rows_per_cond_fn": wl.rows,
                   "cond_fn_per_chain_step": wl.n_obj},
        "ms_per_denoise_step": secs / a.steps / wl.S * 1e3,
        "ms_per_denoise_step_per_pair": secs / a.steps / wl.S / pairs * 1e3,
    }
    if roof:
        line["roofline"] = roof
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(wl)
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    if world == 1 and not a.no_extra:
        other = "2d" if wl.kind == "3d" else "3d"
        del wl
        torch.cuda.empty_cache()
        # the other BASELINE configurations, two timed steps each (not the headline: `value` above is what the driver reads)
        extras = []
        for kind, contraction in ((other, "f32"), ("3d", "bf16"), ("2d", "bf16"), ("3d_ensemble", "bf16")):
            w2 = Workload(kind, {"3d": 32, "2d": 4, "3d_ensemble": 8}[kind], dev, rank, contraction)
            s2, _ = timed_loop(w2, 2, 1, None)
            extras.append({"workload": kind, "dtype": contraction, "samples_per_s": w2.B * w2.pairs * 2 / s2, "ms_per_step": s2 / 2 * 1e3,
                           "ms_per_denoise_step_per_pair": s2 / 2 / w2.S / w2.pairs * 1e3, "cond_fn_per_chain_step": w2.n_obj})
            del w2
            torch.cuda.empty_cache()
        line["extra"] = extras
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
