#!/bin/bash
# timing experiments on the LDS-stream trunk: phase stamps per DGDM_EXP value
for e in ${EXPS:-0 1}; do
  DGDM_EXTRA_FLAGS="-DDGDM_F16_STAMPS -DDGDM_EXP=$e" python -c "import dgdm_amd.build as b; b.build(force=True)" > /dev/null 2>&1
  echo "EXP $e"
  python bench.py --workload 3d --steps 1 --warmup 1 --no-cpu-baseline --contraction f32_f16x3 2>&1 | grep "stamps kind 3" | tail -1
done
