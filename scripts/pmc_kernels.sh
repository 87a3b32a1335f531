#!/bin/bash
# Per-kernel PMC sums for one 3-D bench step (two passes; no --stats / sys-trace with --pmc).  usage: bash scripts/pmc_kernels.sh [extra bench args]
R=$(pwd); OUT=$R/gpurun_out/pmck; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/a -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-2d "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-2d "$@" > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, re
from collections import defaultdict
for d in ("a", "b"):
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("dgdm::", "")[:28]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    names = sorted({c for k in agg for c in agg[k]})
    print("kernel".ljust(30), "n".rjust(5), *[c.replace("SQ_", "")[:16].rjust(17) for c in names])
    for k in sorted(agg, key=lambda k: -agg[k].get(names[0], 0))[:12]:
        print(k.ljust(30), str(len(cnt[k])).rjust(5), *[("%.3e" % agg[k].get(c, 0)).rjust(17) for c in names])
PY
