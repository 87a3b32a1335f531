#!/bin/bash
# Are two builds of the library bit-identical on the guidance gradient?  The tree's libdgdm_hip.so against dgdm_amd/csrc/alt_exp.so.
cp dgdm_amd/csrc/libdgdm_hip.so /tmp/libA.so
trap 'cp /tmp/libA.so dgdm_amd/csrc/libdgdm_hip.so' EXIT      # the shipped library is back in place however the script ends
timeout 200 python3 scripts/dump_grad.py /tmp/gA.npz 2>/dev/null
cp dgdm_amd/csrc/alt_exp.so dgdm_amd/csrc/libdgdm_hip.so
timeout 200 python3 scripts/dump_grad.py /tmp/gB.npz 2>/dev/null
cp /tmp/libA.so dgdm_amd/csrc/libdgdm_hip.so
python3 - <<'PY'
import numpy as np
a, b = np.load("/tmp/gA.npz"), np.load("/tmp/gB.npz")
for k in a.files:
    d = np.abs(a[k].astype(np.float64) - b[k]).max() / np.abs(a[k]).max()
    print(k, "identical" if np.array_equal(a[k], b[k]) else f"DIFFERENT (max rel {d:.2e})")
PY
