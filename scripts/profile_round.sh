#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh r01
# Writes raw output under gpurun_out/prof_<tag>/ and the summaries that get committed under profiles/:
#   <tag>_{3d,2d}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py [--workload 2d] --steps 2 --warmup 1`
#   <tag>_{3d,2d}_pmc_hbm.json       FETCH_SIZE / WRITE_SIZE per kernel, two separate --pmc passes (counter unit: KB)
# PMC passes never carry --stats/sys-trace options (MI355X_MICROARCH.md, HBM section).
set -u
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra"
for wl in 3d 2d; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -o k -- \
      python3 "$R/bench.py" --workload $wl --steps 2 --warmup 1 $COMMON > "$OUT/stats_$wl.log" 2>&1
  f=$(find "$OUT/stats_$wl" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$R/profiles/${TAG}_${wl}_kernel_stats.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${wl}_$c" -o p -- \
        python3 "$R/bench.py" --workload $wl --steps 1 --warmup 0 $COMMON > "$OUT/pmc_${wl}_$c.log" 2>&1
  done
  python3 "$R/scripts/pmc_summary.py" "$OUT/pmc_${wl}_FETCH_SIZE" "$OUT/pmc_${wl}_WRITE_SIZE" > "$R/profiles/${TAG}_${wl}_pmc_hbm.json"
done
cp "$R"/profiles/${TAG}_* "$OUT/" 2>/dev/null
tail -2 "$OUT"/stats_3d.log "$OUT"/stats_2d.log
