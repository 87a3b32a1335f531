#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh r01
# Writes raw output under gpurun_out/prof_<tag>/ and the summaries that get committed under profiles/:
#   <tag>_{3d,2d,3d_bf16}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py [--workload 2d] [--contraction bf16] --steps 2 --warmup 1`
#   <tag>_{3d,2d,3d_bf16}_pmc_hbm.json       FETCH_SIZE / WRITE_SIZE per kernel, two separate --pmc passes (counter unit: KB)
# PMC passes never carry --stats/sys-trace options (MI355X_MICROARCH.md, HBM section).
set -u
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extra --no-2d"
HEAD_ID=${DGDM_HEAD:-unknown}
for cfg in "3d f32" "2d f32" "3d bf16"; do
  set -- $cfg; wl=$1; ct=$2
  tag=$wl; [ "$ct" = bf16 ] && tag=${wl}_bf16
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$tag" -o k -- \
      python3 "$R/bench.py" --workload $wl --contraction $ct --steps 2 --warmup 1 $COMMON > "$OUT/stats_$tag.log" 2>&1
  f=$(find "$OUT/stats_$tag" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$R/profiles/${TAG}_${tag}_kernel_stats.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- \
        python3 "$R/bench.py" --workload $wl --contraction $ct --steps 1 --warmup 0 $COMMON > "$OUT/pmc_${tag}_$c.log" 2>&1
  done
  python3 "$R/scripts/pmc_summary.py" "$OUT/pmc_${tag}_FETCH_SIZE" "$OUT/pmc_${tag}_WRITE_SIZE" "$HEAD_ID" > "$R/profiles/${TAG}_${tag}_pmc_hbm.json"
done
cp "$R"/profiles/${TAG}_* "$OUT/" 2>/dev/null
tail -n 2 "$OUT"/stats_3d.log "$OUT"/stats_2d.log
