#!/bin/bash
# MI355X: kernel + memory-copy timeline of the default 3-D bench (2 steps + 1 warm-up) -> gpurun_out/timeline_<tag>/
set -u
TAG=${1:-t}; shift || true
R=$(pwd)
OUT=$R/gpurun_out/timeline_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o k -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-2d "$@" > "$OUT/run.log" 2>&1
ls "$OUT"
