#!/usr/bin/env python3
"""Summarise two rocprofv3 `--pmc` passes (FETCH_SIZE, WRITE_SIZE; counter unit KB) per kernel:
    python3 scripts/pmc_summary.py <fetch_dir> <write_dir>  > profiles/<tag>_<workload>_pmc_hbm.json
Values are the RAW counters (sum over the dispatch's rows, averaged over launches); bench.py applies nothing to them and
DESIGN.md states how they compare with the algorithmic bytes."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def one(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per_dispatch = defaultdict(float)
    name_of = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                key = (f, row["Dispatch_Id"])
                per_dispatch[key] += float(row["Counter_Value"])
                name_of[key] = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
    agg = defaultdict(lambda: [0, 0.0])
    for key, v in per_dispatch.items():
        a = agg[name_of[key]]
        a[0] += 1
        a[1] += v
    return {k: {"launches": n, "avg_KB": s / n} for k, (n, s) in agg.items() if k.startswith("dgdm::")}


if __name__ == "__main__":
    print(json.dumps({"head": sys.argv[3] if len(sys.argv) > 3 else None, "fetch": one(sys.argv[1], "FETCH_SIZE"), "write": one(sys.argv[2], "WRITE_SIZE")}, indent=1))
