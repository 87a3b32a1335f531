#!/bin/bash
# Same-box comparison of two TREES (e.g. the previous round's, extracted with `git archive <commit> | tar -x -C _r04` and built there):
# three alternating default-workload bench runs each.  usage (through gpurun): bash scripts/ab_rounds.sh _r04
OLD=${1:-_r04}
for i in 1 2 3; do
  (cd $OLD && timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1) > gpurun_out/rounds_old_$i.json
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-2d 2>/dev/null | tail -1 > gpurun_out/rounds_new_$i.json
done
python3 - <<'PY'
import json
for v in ("old", "new"):
    rows = [json.load(open(f"gpurun_out/rounds_{v}_{i}.json")) for i in (1, 2, 3)]
    print(v, "samples/s", [round(r["value"]) for r in rows], "ms/step", [round(r["ms_per_step"], 2) for r in rows],
          "trunk ms/launch", [round(r["roofline"]["avg_launch_ms"], 3) for r in rows], "issued", [round(r["roofline"].get("matrix_pipe_issue_frac", 0), 3) for r in rows])
PY
