#!/bin/bash
# one wave's cycle stamps per phase of trunk_f16l_kernel (3-D); EXPS: extra -D flags per run, e.g. EXPS="- -DDGDM_F16_SLOTS=5" (timing
# experiment with wrong results: no LDS-DMA traffic after a stream's first chunks - measured: the tile's 419 k cycles become 394 k)
for e in ${EXPS:--}; do
  [ "$e" = "-" ] && e=""
  DGDM_EXTRA_FLAGS="-DDGDM_F16_STAMPS $e" python -c "import dgdm_amd.build as b; b.build(force=True)" > /dev/null 2>&1
  echo "flags: $e"
  python bench.py --workload 3d --steps 1 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | grep "stamps kind 3" | tail -1
done
