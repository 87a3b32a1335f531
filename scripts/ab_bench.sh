#!/bin/bash
# Same-box A/B of two builds of the library: the tree's libdgdm_hip.so against dgdm_amd/csrc/alt_exp.so (a variant build copied there),
# three alternating runs of the 3-D bench each.  usage (through gpurun): bash scripts/ab_bench.sh [bench args]
cp dgdm_amd/csrc/libdgdm_hip.so /tmp/libA.so
trap 'cp /tmp/libA.so dgdm_amd/csrc/libdgdm_hip.so' EXIT      # the shipped library is back in place however the script ends
for i in 1 2 3; do
  cp /tmp/libA.so dgdm_amd/csrc/libdgdm_hip.so; timeout 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > gpurun_out/abA_$i.json
  cp dgdm_amd/csrc/alt_exp.so dgdm_amd/csrc/libdgdm_hip.so; timeout 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > gpurun_out/abB_$i.json
done
cp /tmp/libA.so dgdm_amd/csrc/libdgdm_hip.so
python3 - <<'PY'
import json
for v in "AB":
    rows = [json.load(open(f"gpurun_out/ab{v}_{i}.json")) for i in (1, 2, 3)]
    print(v, "samples/s", [round(r["value"]) for r in rows], "ms/step", [round(r["ms_per_step"], 2) for r in rows], "trunk ms/launch", [round(r["roofline"]["avg_launch_ms"], 3) for r in rows])
PY
