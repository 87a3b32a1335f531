#!/bin/bash
# MI355X: the 3-D bench under a list of environment settings (one per argument, e.g. "DGDM_Z64_TILES=1 DGDM_L2C_SPLIT=1"), two runs each;
# prints samples/s, ms per step and the per-stage milliseconds
for cfg in "$@"; do
  for i in 1 2; do echo "$cfg: $(env $cfg timeout 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import json,sys; r=json.loads(sys.stdin.read()); print(round(r["value"]), round(r["ms_per_step"],2), r.get("stage_ms") or "")')"; done
done
