#!/bin/bash
# MI355X: per-kernel time of the default 3-D bench step (rocprofv3 --kernel-trace --stats, 2 steps + 1 warm-up), top rows printed
#   bash scripts/kstats.sh [tag] [extra bench args]
set -u
TAG=${1:-k}; shift || true
R=$(pwd)
OUT=$R/gpurun_out/kstats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-2d "$@" > "$OUT/run.log" 2>&1
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && python3 "$R/scripts/prof_top.py" "$OUT/kernel_stats.csv" 30
tail -n 1 "$OUT/run.log" | cut -c1-400
