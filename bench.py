#!/usr/bin/env python3
"""Benchmark of the guided-sampling hot path (BASELINE.json metric: guided samples/sec over the full DDIM chain,
plus ms per denoise step) on MI355X.

Workload (default, BASELINE configs[2] shape; configs[3] at N = 8): 3-D dynamics-guided sampling.  One *step* of this
benchmark is one batch of `--pairs` independent (object x objective) pairs per GPU, each a complete guided chain of
B = 32 fingers: PointNet++ object tables, then S = 5 x [eps-net, cond_fn over R = 32*45*25 = 36 000 replicated rows with
sub_bs = 512 FPS-start partition, guidance combine, DDIM step].  Every pair has its own synthetic 512-point object, so
nothing is shared between pairs; the tables are rebuilt inside the timed region for every pair.  The FPS start
indices (the torch.randint draws of pointnet2_utils.py:83) are the path's random input: they are drawn INSIDE the timed region, as the
product path draws them - a worker thread makes step k+1's draws (the library's replay of torch's CPU generator, one stream per pair)
while the GPU runs step k - and handed over as host int64 buffers (the boundary's contract).  Every pair has a global index (step, rank, slot) -> its object, objective and its own start
stream are functions of that index only, so the work of a pair does not depend on how many ranks share the batch.
`--workload 2d` runs BASELINE configs[1] (B = 64, G = 360, P = 5, R = 576 000 rows per pair, 100-vertex contours).

Contract: `python bench.py --gpus N --steps K --warmup W`.  One rank per GPU.  Under a launcher (torchrun: RANK / LOCAL_RANK /
WORLD_SIZE in the environment) this process is one rank; without one and N > 1 it starts the N rank processes itself (fresh
interpreters, before anything here touches a GPU) and waits for them.  W untimed steps, exactly K timed steps between
barrier + synchronize, MAX over ranks; rank 0 prints ONE JSON line - the LAST line of stdout, compact (< 4 KB, scalars only:
metric / value / config / roofline / stage_ms / cpu_baseline, and at N = 1 the metric's 2-D half - BASELINE configs[1], 10 steps run
after the 3-D measurement - as value_2d / ms_per_step_2d / roofline_2d_frac / cpu_baseline_2d).  `--extra` additionally runs the secondary workloads and writes them,
with the CPU legs' run lists, to gpurun_out/bench_extra.json; they are never part of the line.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
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
DEFAULT_PAIRS = {"3d": 32, "2d": 4, "3d_ensemble": 8}
STREAM_SEED = 1234


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", choices=["3d", "2d", "3d_ensemble"], default="3d")
    p.add_argument("--pairs", type=int, default=0, help="(object x objective) pairs per GPU per step (default 32 for 3d, 4 for 2d; "
                                                        "3d_ensemble: chains per GPU per step, default 8, each averaging 4 objects' gradients)")
    p.add_argument("--contraction", choices=["f32", "f32_mfma", "bf16", "f32_f16x3"], default="f32",
                   help="arithmetic of the dynamics-trunk contractions: f32 (the parity path, the library default = f32_f16x3: float32 operands "
                        "scaled by exact powers of two and split into two f16 pieces, three f16 MFMAs per product, float32 accumulation), "
                        "f32_mfma (the k-ordered float32 MFMA chain) or "
                        "bf16 (operands ROUNDED to bf16, float32 accumulation)")
    p.add_argument("--cpu-baseline", choices=["sample", "full", "none"], default="full",
                   help="CPU oracle timed beside the GPU number (rank 0, N = 1): full = the SURVEY 8(d) protocol, median of 3 runs of the bounded sample + "
                        "an end-to-end reduced-grid chain as a check of the extrapolation (about 75 s of CPU work; the default) + the CPU legs of every "
                        "--extra workload, sample = ONE run of the bounded sample (about 15 s), none = skip")
    p.add_argument("--no-2d", action="store_true", help="skip the second leg of the default run (BASELINE configs[1], the metric's 2-D half: "
                   "10 steps after the 3-D measurement, reported as value_2d / ms_per_step_2d / roofline_2d_frac / cpu_baseline_2d)")
    p.add_argument("--no-cpu-baseline", action="store_true", help="same as --cpu-baseline none")
    p.add_argument("--extra", action="store_true", help="after the headline measurement also run the secondary workloads (the other BASELINE configs, the "
                   "other contraction modes, the training legs) and write them to --extra-out; never part of the headline line")
    p.add_argument("--no-extra", action="store_true", help="accepted for compatibility (extras are off unless --extra is given)")
    p.add_argument("--extra-out", default=os.path.join(ROOT, "gpurun_out", "bench_extra.json"))
    p.add_argument("--train-only", action="store_true", help="print only the Trainer.step leg of the secondary summary (1 GPU)")
    p.add_argument("--force-group", action="store_true", help="initialise the process group and run its barriers, the gather and the max-over-ranks "
                   "reduction also at world size 1 (test: the whole rank body on RCCL end to end on a 1-GPU box)")
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="collective backend for N > 1: nccl (= RCCL, one GPU per rank; the measured configuration) or gloo (launcher / sharding "
                        "test on a box with fewer GPUs than ranks: ranks share the GPUs round-robin and the final gather goes through host memory)")
    a = p.parse_args()
    if a.no_cpu_baseline:
        a.cpu_baseline = "none"
    return a


# ---------------------------------------------------------------------------------------------------------------- launcher
def spawn_ranks(n: int) -> int:
    """`--gpus N` without a launcher: N fresh rank processes of this script (children, never an exec of this process), one per
    GPU, rendezvous on 127.0.0.1.  The parent touches no GPU; rank 0's stdout is the benchmark's JSON line."""
    have = torch.cuda.device_count()          # counts devices without initialising the runtime
    if have < n and "gloo" not in sys.argv[1:] and "--backend=gloo" not in sys.argv[1:]:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible (RCCL needs one GPU per rank; --backend gloo shares them for a functional test)",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # wait for all; a rank that dies would leave the others blocked in a collective for ever, so the first failure ends the run:
    # the remaining children (exactly the PIDs started above) are terminated and the parent reports the failure
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                print(f"bench.py: rank process {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    q.terminate()
                deadline = time.time() + 10
                for q in live:
                    try:
                        q.wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------------- workload
class Workload:
    def __init__(self, kind, pairs, dev, rank, world, contraction="f32"):
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
        self.rank, self.world = rank, world
        self.objectives = [o for o in synth.OBJECTIVES_12 if o != 'convergence']     # 'convergence' needs the sim-side unguided pass
        self.rows = self.guid.rows

    def pair_ids(self, step):
        """Global indices of this rank's pairs in `step`: the batch of world*pairs pairs is block-partitioned over the ranks."""
        base = (step * self.world + self.rank) * self.pairs
        return [base + i for i in range(self.pairs)]

    def chains(self, step):
        return [(i, self.objectives[g % len(self.objectives)]) for i, g in enumerate(self.pair_ids(step))]

    def objects(self, step):
        """Synthetic objects of this step's pairs (one per gradient chain, a function of the global pair index), on the device."""
        mk = synth.synth_object_3d if self.kind == "3d" else synth.synth_object_2d
        return torch.stack([mk(g * self.n_obj + j, self.N) for g in self.pair_ids(step) for j in range(self.n_obj)]).to(self.dev)

    def draw_buffer(self):
        """A reusable host buffer for one step's FPS start draws (3-D), in the layout the library takes; None for 2-D."""
        if self.kind != "3d":
            return None
        shape = (self.S, self.n_obj, self.pairs, self.guid.starts_per_call) if self.ensemble else (self.S, self.pairs, self.guid.starts_per_call)
        return np.zeros(shape, dtype=np.int64)          # zeros: the pages are touched here, not inside the timed region

    def draw(self, step, out=None, pool=None):
        """The FPS start draws of this step's pairs (pointnet2_utils.py:83): every pair consumes its own generator stream - the stream
        of torch.Generator().manual_seed(f(STREAM_SEED, global pair index)), replayed by the library (sampler.TorchRng) - chain after
        chain as the reference's loops do; independent streams are drawn side by side on `pool`."""
        if self.kind != "3d":
            return None
        streams = [sampler.pair_stream(self.N, self.sub, STREAM_SEED, g) for g in self.pair_ids(step)]
        if self.ensemble:
            return sampler.draw_ensemble_starts(self.guid, self.pairs, self.n_obj, self.S, streams, out=out, pool=pool)
        return sampler.draw_chain_starts(self.guid, self.chains(step), self.S, streams=streams, out=out, pool=pool)

    def run(self, step, objs, predrawn):
        self.guid.set_objects(objs)                      # 3-D: builds the PointNet++ tables of every pair (timed)
        if self.ensemble:                                # chain k averages the gradients of objects 4k..4k+3 (diffusion.py:637-647)
            groups = [list(range(self.n_obj * k, self.n_obj * (k + 1))) for k in range(self.pairs)]
            return sampler.guided_multi_object_groups(self.net, self.guid, self.sched, self.mode, self.noise, groups,
                                                      [o for _, o in self.chains(step)], predrawn=predrawn)
        return sampler.guided_chains(self.net, self.guid, self.sched, self.mode, self.noise, self.chains(step), predrawn=predrawn)


FORCE_GROUP = False


def timed_loop(wl, steps, warmup, dist):
    """Returns (seconds for `steps` steps, last output, mean host seconds per step spent drawing).

    The synthetic objects of every step exist on the device before the clock starts, like a dataset would.  The FPS start draws - the
    path's random input, torch.randint on the CPU generator in the reference (pointnet2_utils.py:83) - are made INSIDE the timed region,
    the way the product path makes them (sampler.StartPlan): a worker thread draws step k+1 while the GPU runs step k.  The first
    timed step's draws start after the clock does (not during the last warm-up step), so all K steps' draws are on the clock and the
    pipeline fill is paid once."""
    from concurrent.futures import ThreadPoolExecutor
    total = warmup + steps
    objs = {k: wl.objects(k) for k in range(total)}
    bufs = [wl.draw_buffer(), wl.draw_buffer()]
    pool = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 8) // max(1, wl.world))))
    drawer = ThreadPoolExecutor(max_workers=1)
    draw_s = {}

    def draw(k):
        t = time.perf_counter()
        r = wl.draw(k, out=bufs[k % 2], pool=pool)
        draw_s[k] = time.perf_counter() - t
        return r

    out, t0, fut, pending = None, None, None, None
    for k in range(total):
        if k == warmup:
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if fut is None:
            fut = drawer.submit(draw, k)
        pre = fut.result()
        fut = drawer.submit(draw, k + 1) if (k + 1 < total and k + 1 != warmup) else None
        out = wl.run(k, objs.pop(k), pre)
        if dist is not None and (dist.get_world_size() > 1 or FORCE_GROUP):
            # the path's only collective: final samples; RCCL on device tensors, or host tensors for the gloo test mode.
            # Asynchronous: RCCL gathers step k's samples on its own stream, behind the chains that made them, while this stream goes on
            # with step k + 1's tables; the previous step's handle is completed here, the last one before the clock stops
            if pending is not None:
                pending.wait()
            pending = gather_pairs(out, wl.pairs * dist.get_world_size(), async_op=True)
    if pending is not None:
        out = pending.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    drawer.shutdown()
    pool.shutdown()
    timed = [draw_s[k] for k in range(warmup, total) if k in draw_s]
    return secs, out, (sum(timed) / len(timed) if timed else 0.0)


# ---------------------------------------------------------------------------------------------------------------- roofline
# what the JSON's `dtype` / roofline.arithmetic say about each contraction mode of the trunk (DESIGN_HISTORY.md 4.1 / 4.6 / 4.10)
DEFAULT_F32_FORM = "f32_f16x3"           # what the library's DGDM_DTYPE_F32 selects (csrc/guidance_api.hip DGDM_DEFAULT_F16X3)
DTYPE_LABEL = {"f32": "f32_split_f16x3", "f32_f16x3": "f32_split_f16x3", "f32_mfma": "f32", "bf16": "bf16"}
ARITHMETIC = {"f32_f16x3": "float32-grade: every float32 product as three f16 MFMA products on two-way-split operands after exact power-of-two scaling "
                           "(per weight matrix, per tile row; float32 accumulation; 1.9e-7 rms of a 256-term contraction vs float64)",
              "f32_mfma": "float32 MFMA (v_mfma_f32_32x32x2_f32), a k-ordered fma chain",
              "bf16": "operands rounded to bf16, float32 accumulation"}
# matrix pipe busy share of the dominant kernel as RECORDED by the latest round's PMC passes (profiles/r0N_pmc_kernels.md, "Derived" table:
# SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)); recorded, not live.  The bf16 kernel keeps its last recording.
PIPE_BUSY_FALLBACK = {("3d", "bf16"): 0.54}


def pipe_busy_recorded(kind, form, kernel):
    import glob
    import re
    if (kind, form) in PIPE_BUSY_FALLBACK:
        return PIPE_BUSY_FALLBACK[(kind, form)], None
    want = "`%s<%s>`" % (kernel, "3" if kind == "3d" else "2")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_kernels.md")), reverse=True):
        for line in open(f):
            m = re.match(r"\|\s*%s\s*\|\s*([0-9.]+)\s*%%" % re.escape(want), line)
            if m:
                return float(m.group(1)) / 100.0, os.path.relpath(f, ROOT)
    return None, None


# algorithmic HBM bytes of ONE cond_fn's trunk launch per (pair, object): the xobj rows (1 KiB per replicated row, 3-D) or nothing of size R (2-D:
# tables only) + the weights once (DESIGN_HISTORY.md 4.1)
HBM_ALGORITHMIC_BYTES = {"3d": 36000 * 32 / 32 * 1024.0 + 7.2e6 / 32, "2d": 17e6 / 4}


def pmc_traffic(workload, contraction, kernel):
    """HBM bytes per launch of the dominant kernel as RECORDED by this round's rocprofv3 PMC passes of this same command
    (scripts/profile_round.sh -> profiles/r03_*_pmc_hbm.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, counter unit KB).
    FETCH_SIZE is the raw counter: on gfx950 it can under-count wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM), so the
    true read traffic lies between 1x and 2x of `fetch_bytes_raw`.  None when no recording exists."""
    tag = workload + ("_bf16" if contraction == "bf16" else "")
    f = next((c for c in (os.path.join(ROOT, "profiles", f"{r}_{tag}_pmc_hbm.json") for r in ("r06", "r05", "r04", "r03")) if os.path.exists(c)), None)
    if f is None:
        return None
    d = json.load(open(f))
    key = next((k for k in d["fetch"] if kernel + "<" in k), None)
    if key is None or key not in d["write"]:
        return None
    fe, wr = d["fetch"][key]["avg_KB"] * 1024.0, d["write"][key]["avg_KB"] * 1024.0
    return {"bytes_per_launch": fe + wr, "fetch_bytes_raw": fe, "write_bytes": wr, "recorded": os.path.relpath(f, ROOT),
            "recorded_at": d.get("head")}


def stage_profile(wl, secs_per_step, contraction):
    """One more (untimed) step with HIP events around every stage's launches on their stream (dgdm_prof_*): the roofline of
    the dominant kernel, the step-level fraction, and the share of each stage in the step."""
    engine.prof_enable(True)
    objs, pre = wl.objects(0), wl.draw(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.run(0, objs, pre)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    st = engine.prof_read_stages()
    engine.prof_enable(False)
    n, ms, flops = st["trunk"]
    if not n:
        return None, None
    # The trunk's arithmetic per mode.  'f32' (default): float32 contractions carried by the f16 matrix pipe - THREE
    # f16 MFMA products are issued per algorithmic float32 product - so the kernel is priced against the f16 dense peak: `frac` with the
    # ALGORITHMIC FLOPs, `matrix_pipe_issue_frac` with the issued ones (3 x); what the algorithmic rate is against the float32-MFMA peak
    # (which the k-ordered chain is bound by) is reported beside it.  'f32_mfma': that chain.  'bf16': operands rounded to bf16.
    form = DEFAULT_F32_FORM if contraction == "f32" else contraction
    kname = {"bf16": "trunk_bf16_kernel", "f32_mfma": "trunk_kernel"}.get(form, "trunk_f16l_kernel")
    issued = {"f32_f16x3": 3}.get(form, 1)
    peak = F32_MFMA_PEAK_TFLOPS if contraction == "f32_mfma" else BF16_MFMA_PEAK_TFLOPS
    alg = flops / (ms * 1e-3) / 1e12
    ach = alg * issued
    need = st["trunk"][2] + st["unet"][2]                  # necessary FLOPs of one step: trunk (real rows) + eps-net (useful MACs)
    tr = pmc_traffic(wl.kind, contraction, kname)
    # `achieved` = ALGORITHMIC TFLOP/s (the float32 contraction FLOPs the path asks for on the real rows / the kernel's time), `peak` = the dense
    # peak of the matrix pipe the kernel runs on, `frac` = their ratio.  For the split float32 form the pipe additionally carries
    # issued_flops_per_algorithmic_flop - 1 redundant piece products per useful one: that utilisation is `matrix_pipe_issue_frac`
    # (issued / peak), NOT `frac`; `frac_algorithmic_vs_f32_peak` prices the same algorithmic rate against the 157.3 TFLOP/s a
    # v_mfma_f32 kernel is bounded by (the figure comparable with the f32_mfma leg and with rounds 1-2).
    # flat scalars only: this object is part of the headline line the driver parses (kept under 4 KB; the prose lives in DESIGN.md §5)
    roof = {"bound": "mfma", "kernel": kname, "achieved": alg, "peak": peak, "unit": "TFLOP/s",
            "frac": alg / peak, "traffic": tr["bytes_per_launch"] if tr else None,
            "traffic_fetch_bytes_raw": tr["fetch_bytes_raw"] if tr else None, "traffic_write_bytes": tr["write_bytes"] if tr else None,
            "traffic_vs_algorithmic": (tr["bytes_per_launch"] / (HBM_ALGORITHMIC_BYTES[wl.kind] * wl.pairs * wl.n_obj)) if tr and wl.kind in HBM_ALGORITHMIC_BYTES else None,
            "traffic_recorded": tr["recorded"] if tr else None,
            "launches": n, "avg_launch_ms": ms / n,
            "algorithmic_flops_per_launch": flops / n, "issued_flops_per_algorithmic_flop": issued,
            "matrix_pipe_issue_frac": ach / peak, "frac_algorithmic_vs_f32_peak": alg / F32_MFMA_PEAK_TFLOPS,
            "pipe_busy_recorded": pipe_busy_recorded(wl.kind, form, kname)[0], "pipe_busy_recorded_in": pipe_busy_recorded(wl.kind, form, kname)[1],
            "arithmetic": form,
            "share_of_step": (ms * 1e-3) / secs_per_step,
            "step_frac": (st["trunk"][2] + st["unet"][2]) / secs_per_step / 1e12 / peak,
            "step_issue_frac": (st["trunk"][2] * issued + st["unet"][2]) / secs_per_step / 1e12 / peak,
            "step_necessary_tflop": need / 1e12}
    shares = {k: round(v[1], 4) for k, v in st.items() if v[0]}          # ms of each stage's launches in one (profiled, untimed) step
    shares["profiled_step_wall"] = round(wall_ms, 4)
    return roof, shares


# ---------------------------------------------------------------------------------------------------------------- CPU legs
def _once(fn):
    t0 = time.perf_counter()
    fn()
    t = time.perf_counter() - t0
    return t, [t]


def _median3(fn):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def host_view(wl):
    """What cpu_baseline needs of a Workload, detached from its device state (the CPU legs run after every GPU leg)."""
    import types
    return types.SimpleNamespace(kind=wl.kind, T=wl.T, S=wl.S, B=wl.B, L=wl.L, G=wl.G, P=wl.P, N=wl.N, sub=getattr(wl, "sub", 0), noise=wl.noise.cpu(),
                                 unet_sd=wl.unet_sd, dyn_sd=wl.dyn_sd)


def cpu_baseline(wl, full=False, median=None):
    """The CPU oracle (a restatement of the reference's as-written dataflow, pinned to the reference by tests/golden) on this
    box's host cores, on a bounded sample of the workload.  Default: ONE run of the sample (about 15 s of CPU work, so
    that the bench command stays mostly GPU time); `full`: median of 3 runs plus one end-to-end chain on a reduced grid as a sanity
    check of the extrapolation."""
    median = full if median is None else median          # median of 3 without the end-to-end check: the 2-D leg of the default line
    med = _median3 if median else _once
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

    def unet():
        with torch.no_grad():
            orc.unet1d_forward(wl.unet_sd, x, ts)
    t_unet, _ = _median3(unet)
    cells = wl.G * wl.P * wl.P
    how = "median of 3 runs" if median else "1 run"
    check = None
    if wl.kind == "3d":
        # one full 512-row sub-batch of cond_fn (fwd + autograd), as generator/diffusion.py:495-498 runs 71 of per step
        s = orc.Setup('point_3d', wl.unet_sd, wl.dyn_sd, so, L, wl.G, wl.P, wl.sub)
        obj = synth.synth_object_3d(0, wl.N)
        ori, pos = orc._pose_grid(s, B, (-1.0, 1.0))
        n = wl.sub

        def sub_batch():
            xr = x.clone().requires_grad_(True)
            with torch.enable_grad():
                pts = orc._pts3d(s, xr).repeat(cells, 1, 1)[:n]
                logits = orc.dyn3d_forward(wl.dyn_sd, pts, ori[:n], pos[:n], ts.repeat(cells)[:n].float() / wl.T,
                                           obj.t().unsqueeze(0).expand(n, -1, -1), None)
                torch.autograd.grad(orc.deltas_to_objective(logits, 'rotate').sum(), xr)
        t_sub, runs = med(sub_batch)
        chain = wl.S * (B * cells * t_sub / n + t_unet)
        sample = (f"{how} of 1 of the {(B * cells + n - 1) // n} sub-batches ({n} of {B * cells} replicated rows) of one cond_fn call "
                  f"(PointNet++ + trunk fwd, autograd bwd) + 1 eps-net forward, extrapolated to {B * cells} rows x {wl.S} steps")
        if full:
            # sanity: a whole guided chain on a reduced grid (G=2, P=2 -> 256 rows per cond_fn = one sub-batch per step)
            Gs, Ps = 2, 2
            s2 = orc.Setup('point_3d', wl.unet_sd, wl.dyn_sd, so, L, Gs, Ps, wl.sub)
            torch.manual_seed(0)
            t0 = time.perf_counter()
            orc.guided_sample(s2, x, obj, 'rotate')
            t_chain = time.perf_counter() - t0
            rows_small = B * Gs * Ps * Ps
            check = {"what": f"one full guided chain end to end at G={Gs}, P={Ps} ({rows_small} rows per cond_fn, {wl.S} steps)",
                     "seconds": t_chain, "predicted_from_sample_s": wl.S * (rows_small * t_sub / n + t_unet)}
    else:
        s = orc.Setup('point', wl.unet_sd, wl.dyn_sd, so, L, wl.G, wl.P)
        obj = synth.synth_object_2d(0, wl.N)
        Bs = 4                                          # 4 of 64 fingers against the full 9000-cell grid (36 000 rows)
        t_c, runs = med(lambda: orc.cond_fn(s, x[:Bs], ts[:Bs], 'rotate', obj))
        chain = wl.S * (t_c * (B / Bs) + t_unet)
        sample = (f"{how} of cond_fn on {Bs} of {B} fingers x all {cells} cells (36000 rows) + 1 eps-net forward, "
                  f"extrapolated x{B // Bs} x{wl.S} steps")
        if full:
            t0 = time.perf_counter()
            orc.guided_sample(s, x[:Bs], obj, 'rotate')     # a whole chain for those 4 fingers at the full grid
            t_chain = time.perf_counter() - t0
            check = {"what": f"one full guided chain end to end for {Bs} fingers at the full grid ({wl.S} steps)", "seconds": t_chain,
                     "predicted_from_sample_s": wl.S * (t_c + t_unet)}
    out = {"value": B / chain, "unit": "samples/s", "cores": cores, "kind": "port", "sample": sample,
           "ms_per_denoise_step": chain / wl.S * 1e3, "cpu_seconds": sum(runs)}
    if check:               # the reduced-grid end-to-end chain beside what the sample predicts for it
        out["check_chain_s"], out["check_predicted_s"] = check["seconds"], check["predicted_from_sample_s"]
    detail = {"runs_s": runs, "end_to_end_check": check}
    return out, detail


def config0(dev, cpu=True):
    """BASELINE configs[0]: 2-D unconditional sampling, B = 4, L = 14, T = S = 1000 (generator/diffusion.py:249-256 with the
    parser defaults dynamics/parser.py:29,31) - the reference's CPU-runnable plumbing case: the CPU oracle chain (median of 3)
    next to the HIP unguided loop on the same noise."""
    from oracle import dgdm_oracle as orc
    torch.set_num_threads(min(os.cpu_count() or 1, 32))      # same policy as cpu_baseline (tiny ops get slower with more threads)
    B, L, T = 4, 14, 1000
    usd = synth.synth_state_dict(synth.unet_spec(), 11)
    noise = synth.synth_noise(0, B, L)
    so = orc.DDIM(T)
    so.set_timesteps(T)
    s = orc.Setup('point', usd, None, so, L, 1, 1)
    if cpu:
        cpu_s, runs = _median3(lambda: orc.unguided_sample(s, noise))
        return {"cpu_oracle_chain_s": cpu_s, "cpu_runs_s": runs, "cpu_cores": torch.get_num_threads(), "cpu_samples_per_s": B / cpu_s}
    net = engine.Unet1d(usd)
    sch = DDIMScheduler(num_train_timesteps=T)
    sch.set_timesteps(T)
    x = noise.to(dev)
    sampler.unguided_sample(net, sch, x)                # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampler.unguided_sample(net, sch, x)
    torch.cuda.synchronize()
    hip_s = time.perf_counter() - t0
    return {"workload": "2d_unconditional (BASELINE configs[0]: B=4, L=14, T=S=1000)", "dtype": "f32",
            "hip_chain_s": hip_s, "hip_samples_per_s": B / hip_s, "hip_ms_per_denoise_step": hip_s / T * 1e3}


def sweep_leg(dev):
    """The workload of the reference's shipped generator/guided_sample_3d.sh: ONE set of 6 objects, 16 fingers, and on it the whole
    validation sweep of generator/diffusion.py:307-339 - 12 objectives, each a multi-object chain (not for 'convergence') and the 6
    per-object chains, 5 denoise steps each.  Unlike the headline workload (a fresh object per pair) the objects are reused 71 times,
    which is what the per-object tables are for; the second figure forces the per-step gather kernels (test hook mode 3) to show what
    the embedding table X[s1][q] (built once the objects have served more than 5 calls) is worth."""
    B, G, P, L, N, sub, T, S, n_obj = 16, 45, 5, 42, 512, 512, 15, 5, 6
    net = engine.Unet1d(synth.synth_state_dict(synth.unet_spec(), 11))
    dyn = engine.Dynamics(3, synth.synth_state_dict(synth.dyn3d_spec(L), 33), L)
    sched = DDIMScheduler(num_train_timesteps=T)
    sched.set_timesteps(S)
    noise = synth.synth_noise(0, B, L).to(dev)
    objs = torch.stack([synth.synth_object_3d(900 + i, N) for i in range(n_obj)]).to(dev)
    out = {}
    for label, mode in (("with_embedding_tables", 0), ("gather_kernels_only", 3)):
        times = []
        for rep in range(2):                       # first repetition warms the allocator up
            guid = engine.Guidance(dyn, B, G, P, (-1.0, 1.0), n_obj, T, N, sub, max_objects=n_obj)
            guid.debug_fps_path(mode)
            torch.manual_seed(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            guid.set_objects(objs)
            ug = sampler.unguided_sample(net, sched, noise)
            n = 0
            for o in synth.OBJECTIVES_12:
                if o != 'convergence':
                    sampler.guided_multi_object(net, guid, sched, 'point_3d', noise, list(range(n_obj)), o)
                    n += B
                sampler.guided_chains(net, guid, sched, 'point_3d', noise, [(i, o) for i in range(n_obj)], unguided=ug)
                n += B * n_obj
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            del guid
        out[label] = {"seconds": times[-1], "samples_per_s": n / times[-1]}
    return {"workload": "3d_sweep (generator/guided_sample_3d.sh: 6 objects x 12 objectives, B=16, G=45, P=5, sub_bs=512, T=15/S=5; objects reused)",
            "dtype": "f32", "samples": n, **out}


def train_leg(dev, with_cpu=True):
    """Trainer.step of the 2-D dynamics model (dynamics/trainer.py:53-103, configuration of
    dynamics/train_dynamics_2d.sh: batch_size 128, ctrlpts_dim 14, 100-vertex objects, T = 15) on one batch of 128 samples x the 9000
    pose cells of the 2-D grid = 1 152 000 rows.  Timed: the C-ABI step (forward with batch statistics, loss, backward, Adam) on
    device-resident inputs, HIP events on the launch stream; the reference's CPU-generator draws (16 M normals per step) are host
    work outside it and reported beside it.  FLOPs = 2 x rows x weights for the forward, twice that for the backward (no input
    gradient below the three first layers)."""
    import argparse
    import ctypes as C
    from dgdm_amd._lib import check, dptr, lib, stream_ptr
    from dgdm_amd.dynamics.trainer import Trainer
    L, nv, T, rows = 14, 100, 15, 128 * 9000
    args = argparse.Namespace(use_sub_batch=False, sub_bs=1024, grid_size=360, learning_rate=1e-4, weight_decay=0.0, num_epochs=100,
                              checkpoint_path=None, fingers_3d=False, ctrlpts_dim=L, object_max_num_vertices=nv, num_timesteps_per_batch=1,
                              num_inference_steps=5, num_train_timesteps=T)
    sd = synth.synth_state_dict(synth.dyn2d_spec(L, 2 * nv), 41)
    import contextlib, io
    tr = Trainer(args)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.create_model(state_dict=sd)
    g = torch.Generator().manual_seed(3)
    data = [torch.rand(shape, generator=g) * 2 - 1 for shape in ((rows, L), (rows, 3), (rows, 1), (rows, 2), (rows, 2 * nv))]
    data = [d.to(dev) for d in data]
    t0 = time.perf_counter()
    inp = tr._inputs(*data)
    torch.cuda.synchronize()
    host_s = time.perf_counter() - t0
    c, nz, sa, sb, tt, o, p, ob, sc, _ = inp
    pred = torch.empty((rows, 3), device=dev)
    loss = C.c_float()

    def step():
        tr._hint(0, rows, 9000)          # what dynamics/main.py tells the trainer: 15 distinct timesteps, 9000 rows per sample's object
        check(lib().dgdm_trainer2d_step(tr._h, dptr(c), dptr(nz), dptr(sa), dptr(sb), dptr(tt), dptr(o), dptr(p), dptr(ob), dptr(sc), rows, 1e-4, 1,
                                        dptr(pred), C.byref(loss), stream_ptr()))
    step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 3
    ev[0].record()
    for _ in range(n):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    secs = ev[0].elapsed_time(ev[1]) / 1e3 / n
    # FLOPs executed: the gripper encoder, the trunk and the head on every row; the time and object encoders on their 15 / 128 distinct
    # inputs (exact de-duplication, dgdm_trainer2d_set_groups) - their share is negligible and not counted
    K = [L, 256, 795] + [256] * 7
    flops = rows * (3 * 2 * 256 * sum(K) - 2 * 256 * (L + 27) + 3 * 2 * 3 * 256)
    as_written = rows * (3 * 2 * 256 * sum([L, 256, 2 * nv, 256, 128, 256, 795] + [256] * 7) - 2 * 256 * (L + 2 * nv + 128 + 27) + 3 * 2 * 3 * 256)
    # the whole Trainer.step as dynamics/main.py calls it: the reference's CPU-generator draws (made one step ahead by a worker thread),
    # their upload, and the GPU step - wall clock over a few steps
    tr.step(*data, rows_per_sample=9000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nw = 4
    for _ in range(nw):
        tr.step(*data, rows_per_sample=9000)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / nw
    out = {"workload": "train2d (Trainer.step, dynamics/train_dynamics_2d.sh: 128 samples x 9000 pose cells = 1152000 rows, L=14, 100-vertex objects, T=15)",
           "dtype": "f32", "rows_per_s": rows / secs, "ms_per_step": secs * 1e3, "loss": float(loss.value),
           "host_draws_and_uploads_ms": host_s * 1e3, "wall_ms_per_step_python_api": wall * 1e3, "rows_per_s_python_api": rows / wall,
           "roofline": {"bound": "mfma", "achieved": flops / secs / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / secs / 157.3e12,
                        "traffic": None, "flops_per_step": flops, "as_written_flops_per_step": as_written,
                        "note": "whole step (GEMM launches + reductions + Adam), float32 MFMA peak; FLOPs = executed (time / object encoders de-duplicated)"}}
    del tr, data, inp, c, nz, sa, sb, tt, o, p, ob, sc, pred
    torch.cuda.empty_cache()
    if with_cpu:
        out["cpu_baseline"] = train_leg_cpu()
    return out


def train_leg_cpu():
    L, nv, T = 14, 100, 15
    sd = synth.synth_state_dict(synth.dyn2d_spec(L, 2 * nv), 41)
    out = {}
    if True:
        from oracle import dgdm_oracle as orc
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        rs = 128 * 36
        g = torch.Generator().manual_seed(4)
        small = [torch.rand(shape, generator=g) * 2 - 1 for shape in ((rs, L), (rs, 3), (rs, 1), (rs, 2), (rs, 2 * nv))]
        ot = orc.Trainer2D(sd, T, 1e-4)
        t_c, runs = _median3(lambda: ot.step(*small))
        out = {"value": rs / t_c, "unit": "rows/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"median of 3 Trainer2D.step (torch CPU autograd) on {rs} rows (128 samples x 36 pose cells)", "runs_s": runs}
    return out


def unet_train_leg(dev):
    """Diffusion.training_step of the eps-net (generator/diffusion.py:126-177, flags of generator/train_diffusion_3d.sh:
    batch 1024, L = 42) - one C-ABI step (noisy input, forward, MSE loss, backward with every weight gradient, Adam) on device-resident
    inputs, HIP events.  FLOPs = 3 x 2 x 82.0 MMAC per sample (forward + input gradients + weight gradients)."""
    B, L, T = 1024, 42, 15
    tr = engine.UnetTrainer(synth.synth_state_dict(synth.unet_spec(), 11), L)
    g = torch.Generator().manual_seed(5)
    x0 = (torch.rand((B, L, 1), generator=g) * 2 - 1).to(dev)
    noise = torch.randn((B, L, 1), generator=g).to(dev)
    ts = torch.randint(0, T, (B,), generator=g).to(dev)
    ac = DDIMScheduler(num_train_timesteps=T).alphas_cumprod.to(dev)[ts]
    sa, sb = ac ** 0.5, (1 - ac) ** 0.5
    for _ in range(2):
        tr.step(x0, noise, sa, sb, ts, 1e-4, want_loss=False)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 10
    ev[0].record()
    for _ in range(n):
        tr.step(x0, noise, sa, sb, ts, 1e-4, want_loss=False)
    ev[1].record()
    torch.cuda.synchronize()
    secs = ev[0].elapsed_time(ev[1]) / 1e3 / n
    flops = B * 3 * 2 * 82.0e6
    return {"workload": "train_eps_net (Diffusion.training_step, generator/train_diffusion_3d.sh: batch 1024, L=42, T=15)", "dtype": "f32",
            "samples_per_s": B / secs, "ms_per_step": secs * 1e3,
            "roofline": {"bound": "mfma", "achieved": flops / secs / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / secs / 1e12 / F32_MFMA_PEAK_TFLOPS,
                         "traffic": None, "note": "whole step (window GEMMs on the float32 MFMA + GroupNorm/Mish passes + reductions + Adam); useful FLOPs "
                                                  "(padding rows of the window layout not counted)"}}


def unet_train_leg_cpu():
    from oracle import dgdm_oracle as orc
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    B, L = 64, 42
    ot = orc.UnetTrainer(synth.synth_state_dict(synth.unet_spec(), 11), 15, L, 1e-4)
    x0 = torch.rand((B, L, 1)) * 2 - 1
    t_c, runs = _median3(lambda: ot.step(x0))
    return {"value": B / t_c, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of 3 UnetTrainer.step (torch CPU autograd + Adam + EMA) on {B} samples, L={L}", "runs_s": runs}


def train3d_leg(dev, rows=512):
    """Trainer.step of the 3-D dynamics model (dynamics/trainer.py:53-103 with --fingers_3d: PointNet++ in training mode
    as written) on one slice of `rows` (control points, pose, 512-point cloud) rows; dynamics/train_dynamics_3d.sh slices at --sub_bs=2048
    (139 GB of activations: fits one MI355X; the leg runs 512 rows = 35 GB to keep the default bench short).  FLOPs = 3 x 2 x 552.7 MMAC
    per row."""
    import argparse
    import contextlib
    import io
    from dgdm_amd.dynamics.trainer import Trainer
    L, N, T = 42, 512, 15
    args = argparse.Namespace(use_sub_batch=False, sub_bs=rows, grid_size=45, learning_rate=1e-4, weight_decay=0.0, num_epochs=100, checkpoint_path=None,
                              fingers_3d=True, ctrlpts_dim=L, object_max_num_vertices=N, num_timesteps_per_batch=1, num_inference_steps=5, num_train_timesteps=T)
    tr = Trainer(args)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.create_model(state_dict=synth.synth_state_dict(synth.dyn3d_spec(L), 33))
    g = torch.Generator().manual_seed(6)
    ctrl = torch.rand((rows, 3, L), generator=g) * 2 - 1
    obj = torch.stack([synth.synth_object_3d(700 + i % 4, N) for i in range(rows)]).permute(0, 2, 1).contiguous()
    ori, pos, score = torch.rand((rows, 1), generator=g) * 2 - 1, torch.rand((rows, 2), generator=g) * 2 - 1, torch.randn((rows, 3), generator=g)
    data = [d.to(dev) for d in (ctrl, score, ori, pos, obj)]
    tr.step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        tr.step(*data)
    torch.cuda.synchronize()
    secs = (time.perf_counter() - t0) / n
    flops = rows * 3 * 2 * 552.7e6
    out = {"workload": f"train3d (Trainer.step --fingers_3d, one slice of {rows} rows x 512-point clouds; dynamics/train_dynamics_3d.sh uses --sub_bs=2048)",
           "dtype": "f32", "rows_per_s": rows / secs, "ms_per_step_python_api": secs * 1e3,
           "roofline": {"bound": "mfma", "achieved": flops / secs / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / secs / 1e12 / F32_MFMA_PEAK_TFLOPS,
                        "traffic": None, "note": "whole step through the Python API incl. the host's draws and uploads; as-written FLOPs (PointNet++ on every row)"}}
    del tr, data
    torch.cuda.empty_cache()
    return out


def train3d_leg_cpu():
    from oracle import dgdm_oracle as orc
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    rows, L = 8, 42
    ot = orc.Trainer3D(synth.synth_state_dict(synth.dyn3d_spec(L), 33), 15, 1e-4)
    g = torch.Generator().manual_seed(6)
    ctrl = torch.rand((rows, 3, L), generator=g) * 2 - 1
    obj = torch.stack([synth.synth_object_3d(700 + i % 4, 512) for i in range(rows)]).permute(0, 2, 1).contiguous()
    ori, pos, score = torch.rand((rows, 1), generator=g) * 2 - 1, torch.rand((rows, 2), generator=g) * 2 - 1, torch.randn((rows, 3), generator=g)
    t_c, runs = _median3(lambda: ot.step(ctrl, score, ori, pos, obj, ot.draw(ctrl), orc.StartLog()))
    return {"value": rows / t_c, "unit": "rows/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of 3 Trainer3D.step (torch CPU autograd, PointNet++ in training mode) on {rows} rows", "runs_s": runs}


# ---------------------------------------------------------------------------------------------------------------- main
def main():
    a = parse()
    global FORCE_GROUP
    FORCE_GROUP = a.force_group
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))                   # the parent never touches a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    if rank != 0:
        # only rank 0 reports: whatever a library on another rank writes to stdout (under torchrun all ranks share it) must not follow the headline
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    dist = None
    if world > 1 or a.force_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if a.backend == "gloo":
            local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dist.init_process_group(a.backend, rank=rank, world_size=world)
        # ranks share the host: keep torch's CPU pools (input synthesis, FPS start draws) from oversubscribing it
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // (2 * world)))
    _lib.device_init(local)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if os.environ.get("DGDM_BENCH_TEST_DIE_RANK") == str(rank):      # test hook (tests/test_gpu_api.py): this rank dies after the rendezvous
        os._exit(3)
    if a.train_only:
        print(json.dumps(train_leg(dev, a.cpu_baseline != "none")))
        return
    pairs = a.pairs or DEFAULT_PAIRS[a.workload]
    wl = Workload(a.workload, pairs, dev, rank, world, a.contraction)

    engine.prof_enable(False)
    secs, _, draw_secs = timed_loop(wl, a.steps, a.warmup, dist)
    if dist is not None:
        tmax = torch.tensor([secs], device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        secs = float(tmax.item())
    if rank != 0:
        dist.destroy_process_group()
        return
    roof, shares = stage_profile(wl, secs / a.steps, a.contraction)
    meas = {"samples": wl.B * pairs * world * a.steps, "secs": secs, "world": world, "steps": a.steps, "warmup": a.warmup, "contraction": a.contraction,
            "workload": a.workload, "pairs": pairs, "B": wl.B, "S": wl.S, "rows": wl.rows, "n_obj": wl.n_obj, "draw_secs": draw_secs,
            "gloo": world > 1 and a.backend == "gloo", "unet_form": "%s%s" % ((lambda f: (f[0], " batched" if f[1] else ""))(wl.net.effective_form(wl.B * pairs, wl.L)))}
    cpu, detail = None, {}
    second = None
    if world == 1 and a.workload == "3d" and not a.no_2d and not a.force_group:
        # the metric's 2-D half (BASELINE configs[1]) in the same line: 10 timed steps after 2 warm-up steps, its trunk's roofline fraction
        # from HIP events, and (below, with the CPU legs) the oracle on a bounded sample of it
        w2 = Workload("2d", DEFAULT_PAIRS["2d"], dev, rank, world, a.contraction)
        n2 = 10
        s2, _, _ = timed_loop(w2, n2, 2, None)
        r2, sh2 = stage_profile(w2, s2 / n2, a.contraction)
        second = {"value_2d": w2.B * w2.pairs * n2 / s2, "ms_per_step_2d": s2 / n2 * 1e3, "ms_per_denoise_step_2d": s2 / n2 / w2.S * 1e3,
                  "steps_2d": n2, "pairs_per_step_2d": w2.pairs, "fingers_per_pair_2d": w2.B, "rows_per_cond_fn_2d": w2.rows,
                  "roofline_2d_frac": r2["frac"] if r2 else None, "roofline_2d_achieved_tflops": r2["achieved"] if r2 else None,
                  "roofline_2d_avg_launch_ms": r2["avg_launch_ms"] if r2 else None,
                  "roofline_2d_issue_frac": r2["matrix_pipe_issue_frac"] if r2 else None, "trunk_share_of_step_2d": r2["share_of_step"] if r2 else None}
        detail["second_leg_2d"] = {"roofline": r2, "stage_ms": sh2}
        hv2 = host_view(w2)
        del w2
        torch.cuda.empty_cache()
    if world == 1 and a.cpu_baseline != "none":
        cpu, detail["cpu_baseline"] = cpu_baseline(host_view(wl), full=a.cpu_baseline == "full")
        if second is not None:
            c2, detail["cpu_baseline_2d"] = cpu_baseline(hv2, full=False, median=a.cpu_baseline == "full")
            second["cpu_baseline_2d"], second["cpu_baseline_2d_cores"] = c2["value"], c2["cores"]
    if world == 1 and a.extra:
        kind = wl.kind
        del wl
        torch.cuda.empty_cache()
        detail["extra"] = extra_legs(dev, kind, rank, world, a.cpu_baseline == "full")
    if detail:
        try:
            os.makedirs(os.path.dirname(a.extra_out), exist_ok=True)
            with open(a.extra_out, "w") as f:
                json.dump({"headline": headline(meas, roof, shares, cpu, second), **detail}, f, indent=1)
            print(f"bench.py: details in {a.extra_out}", file=sys.stderr)
        except OSError as e:                                 # a read-only tree must not cost the headline
            print(f"bench.py: could not write {a.extra_out}: {e}", file=sys.stderr)
    # the LAST stdout line is the record the driver parses: compact (< 4 KB), scalars only - printed after the process group is gone, so that
    # nothing a collective library says at teardown can follow it
    if dist is not None:
        dist.destroy_process_group()
    # librccl announces itself ("RCCL version ... Librccl path ...") through C stdio, which on a pipe sits in libc's buffer until the process
    # exits - i.e. it would land BEHIND a line printed from Python: flush libc's buffers first
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(json.dumps(headline(meas, roof, shares, cpu, second)), flush=True)


def headline(m, roof, shares, cpu, second=None):
    """The one JSON line of the bench contract, from plain numbers (tests/test_host_logic.py builds it from canned ones): kept far below
    4 KB so that a tail-capturing driver always holds all of it.  Prose about what the figures mean is in DESIGN.md §5, not here."""
    secs, steps = m["secs"], m["steps"]
    line = {"metric": "guided samples/sec (full DDIM chain)", "value": m["samples"] / secs, "unit": "samples/s", "n_gpus": m["world"], "steps": steps,
            "warmup": m["warmup"], "ms_per_step": secs / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[m["contraction"]], "data": "synthetic",
            "config": {"workload": WORKLOAD_NAME[m["workload"]], "contraction_flag": m["contraction"], "pairs_per_gpu_per_step": m["pairs"],
                       "fingers_per_pair": m["B"], "denoise_steps": m["S"], "rows_per_cond_fn": m["rows"], "cond_fn_per_chain_step": m["n_obj"],
                       "eps_net_form": m.get("unet_form"), "parallelism": f"pairs-sharded x{m['world']}"},
            "ms_per_denoise_step": secs / steps / m["S"] * 1e3,
            "ms_per_denoise_step_per_pair": secs / steps / m["S"] / m["pairs"] * 1e3,
            "host_draw_ms_per_step": m["draw_secs"] * 1e3}
    if m.get("gloo"):
        line["backend_note"] = "gloo test mode: ranks share GPUs, not a scaling measurement"
    if roof:
        line["roofline"] = roof
        line["stage_ms"] = shares
    if cpu:
        line["cpu_baseline"] = cpu
        line["speedup_vs_cpu_baseline"] = line["value"] / cpu["value"]
    if second:
        line.update(second)           # flat scalars: the 2-D half of the metric (BASELINE configs[1]), WORKLOAD_NAME["2d"]
        line["config"]["workload_2d"] = WORKLOAD_NAME["2d"]
    return _round(line)


def _round(o):
    """Seven significant digits for every float of the line (shorter record; nothing here is known better than that)."""
    if isinstance(o, float):
        return float(f"{o:.7g}")
    if isinstance(o, dict):
        return {k: _round(v) for k, v in o.items()}
    return o


WORKLOAD_NAME = {"3d": "3d_guided BASELINE configs[2] (configs[3] at 8 GPUs): B=32 G=45 P=5 R=36000 sub_bs=512 N=512 T=15/S=5, fresh object per pair",
                 "2d": "2d_guided BASELINE configs[1]: B=64 G=360 P=5 R=576000 100-vertex objects T=15/S=5, fresh object per pair",
                 "3d_ensemble": "3d_guided_ensemble BASELINE configs[4]: 4 cond_fn per chain-step averaged (guided_sample_multi_object), B=32 G=45 P=5 R=36000 T=15/S=5"}


def extra_legs(dev, kind, rank, world, with_cpu):
    """`--extra`: the other BASELINE configurations, the other contraction modes and the training legs (not the headline).  Every GPU leg
    first and back to back, every CPU leg after them."""
    other = "2d" if kind == "3d" else "3d"
    c0, tl, ul, t3 = config0(dev, cpu=False), train_leg(dev, False), unet_train_leg(dev), train3d_leg(dev)
    extras = [c0, sweep_leg(dev), tl, ul, t3]
    cpu_jobs = []
    if with_cpu:
        cpu_jobs += [lambda: c0.update(config0(dev, cpu=True)), lambda: tl.update(cpu_baseline=train_leg_cpu()),
                     lambda: ul.update(cpu_baseline=unet_train_leg_cpu()), lambda: t3.update(cpu_baseline=train3d_leg_cpu())]
    for k2, contraction in ((other, "f32"), ("3d", "f32_mfma"), ("3d", "bf16"), ("2d", "bf16"), ("3d_ensemble", "bf16")):
        w2 = Workload(k2, DEFAULT_PAIRS[k2], dev, rank, world, contraction)
        ns = 4
        s2, _, d2 = timed_loop(w2, ns, 1, None)
        e = {"workload": k2, "dtype": DTYPE_LABEL[contraction], "contraction_flag": contraction, "samples_per_s": w2.B * w2.pairs * ns / s2, "ms_per_step": s2 / ns * 1e3,
             "ms_per_denoise_step_per_pair": s2 / ns / w2.S / w2.pairs * 1e3, "cond_fn_per_chain_step": w2.n_obj, "host_draw_ms_per_step": d2 * 1e3}
        r2, sh2 = stage_profile(w2, s2 / ns, contraction)
        if r2:
            e["roofline"], e["stage_ms"] = r2, sh2
        if contraction == "f32" and k2 == other and with_cpu:
            hv2 = host_view(w2)
            cpu_jobs.append(lambda e=e, hv2=hv2: e.update(cpu_baseline=cpu_baseline(hv2, full=True)[0]))
        extras.append(e)
        del w2
        torch.cuda.empty_cache()
    for job in cpu_jobs:
        job()
    return extras


if __name__ == "__main__":
    main()
