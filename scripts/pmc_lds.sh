#!/bin/bash
# LDS bank-conflict share per kernel: SQ_LDS_BANK_CONFLICT (extra LDS-array cycles) / SQ_LDS_IDX_ACTIVE (all LDS-array cycles), one bench step.
R=$(pwd); OUT=$R/gpurun_out/pmclds; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/a -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > $OUT/a.log 2>&1
python3 - <<PY
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float))
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("dgdm::", "")[:30]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:8]:
    a, c, n = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_INSTS_LDS", 0)
    print(f"{k:32s} LDS instr {n:.3e}  idx_active {a:.3e}  bank_conflict {c:.3e}  conflict share {c / a if a else 0:.3f}")
PY
