"""Builds dgdm_amd/csrc/libdgdm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The library is kept in-tree so that it travels with the source snapshot to the GPU box."""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libdgdm_hip.so")
SOURCES = ["host_util.hip", "smallnet.hip", "unet.hip", "trunk.hip", "trunk_bf16.hip", "trunk_f16l.hip", "pointnet.hip", "pointnet64.hip", "models_api.hip", "guidance_api.hip", "decode.hip", "debug.hip", "train2d.hip", "unet_train.hip", "train3d.hip", "torch_rng.hip"]
# unet.hip: its block functions as real calls cost 200 VGPRs and a register save/restore through scratch at every call (152 MB of
# scratch writes per 1024-sample launch in the round-2 PMC pass); fully inlined the kernel needs 126 VGPRs
PER_FILE_FLAGS = {"unet.hip": ["-mllvm", "-amdgpu-function-calls=false"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("DGDM_EXTRA_FLAGS", "").split()


INFO = LIB + ".json"      # provenance of the shipped library: content hashes of what it was built from (travels with the .so)


def _sha(paths, extra=()) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    for e in extra:
        h.update(str(e).encode())
    return h.hexdigest()


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(HERE, "..", "include", "dgdm_hip.h")]


def source_hashes() -> dict:
    """Per object file: the hash of its source, every header and its flags - what decides whether it is rebuilt (content, not mtimes: a
    checkout or a copy to another box changes mtimes, not sources)."""
    hd = _headers()
    return {src: _sha([os.path.join(CSRC, src)] + hd, FLAGS + PER_FILE_FLAGS.get(src, [])) for src in SOURCES}


def build_info() -> dict:
    """What the library on disk was built from ({} when there is none) and whether that is what the tree holds now."""
    if not (os.path.exists(LIB) and os.path.exists(INFO)):
        return {}
    with open(INFO) as f:
        info = json.load(f)
    # current = built from the sources the tree holds now AND still the file that build wrote (an experiment script that copies a
    # variant library over it, or a partial build, must not be certified by the manifest of the last good one)
    info["lib_intact"] = info.get("lib_sha256") == _sha([LIB])
    info["current"] = info.get("sources") == source_hashes() and info["lib_intact"]
    return info


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    want = source_hashes()
    have = {}
    if os.path.exists(INFO):
        with open(INFO) as f:
            have = json.load(f).get("sources", {})
    objs, jobs, rebuilt = [], [], []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, "build", src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or have.get(src) != want[src]:
            jobs.append([hipcc] + FLAGS + PER_FILE_FLAGS.get(src, []) + ["-c", s, "-o", o])
            rebuilt.append(src)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(run, jobs):
            if verbose and err.strip():
                print(err)
    intact = os.path.exists(LIB) and os.path.exists(INFO) and json.load(open(INFO)).get("lib_sha256") == _sha([LIB])
    linked = bool(force or jobs or not intact)
    if linked:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        with open(INFO, "w") as f:
            json.dump({"sources": want, "flags": FLAGS, "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "recompiled": rebuilt,
                       "lib_sha256": _sha([LIB])}, f, indent=1)
    build.last = {"mode": "rebuilt" if linked else "up to date", "recompiled": rebuilt}
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
