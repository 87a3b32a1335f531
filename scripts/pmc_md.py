"""profiles/<tag>_pmc_kernels.md from the two passes of scripts/pmc_kernels.sh (gpurun_out/pmck/{a,b}): the raw per-kernel sums and the
derived figures bench.py reads (`pipe_busy_recorded`: first percentage of a kernel's row in the Derived table).
usage: python3 scripts/pmc_md.py r06 > profiles/r06_pmc_kernels.md"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
agg, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(set)
raw = []
for d in ("a", "b"):
    a2, c2 = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmck", d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("dgdm::", "")[:28]
            a2[k][r["Counter_Name"]] += float(r["Counter_Value"])
            c2[k].add(r["Dispatch_Id"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for k in a2 for c in a2[k]})
    raw.append("kernel".ljust(30) + " " + "n".rjust(5) + " " + " ".join(c.replace("SQ_", "")[:16].rjust(17) for c in names))
    for k in sorted(a2, key=lambda k: -a2[k].get(names[0], 0))[:14]:
        raw.append(k.ljust(30) + " " + str(len(c2[k])).rjust(5) + " " + " ".join(("%.3e" % a2[k].get(c, 0)).rjust(17) for c in names))
print(f"# Round {tag[1:].lstrip('0')}: per-kernel PMC sums (`scripts/pmc_kernels.sh`, one profiled 3-D bench step; two `--pmc` passes, no trace options beside `--kernel-trace`)\n")
print("Counters are sums over the launches the pass saw (`n`).  `SQ_WAVE_CYCLES`, `SQ_WAIT_*`, `SQ_ACTIVE_INST_*` count quad-cycles; `SQ_VALU_MFMA_BUSY_CYCLES` counts cycles "
      "(32 per `v_mfma_f32_32x32x16_f16`, 16 per `v_mfma_f32_16x16x32_f16`); `GRBM_GUI_ACTIVE` is summed over the 8 XCDs.  Column names are cut to 16 characters by the script.\n")
print("## Derived\n")
print("| kernel | MFMA pipe busy = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8) | other VALU per MFMA | VALU per LDS instruction | WAIT_ANY / WAVE_CYCLES | WAIT_INST_ANY / WAVE_CYCLES | LDS instruction cycles (4 per instruction) + bank-conflict cycles / (GUI_ACTIVE / 8 x 256 CUs) |")
print("|---|---|---|---|---|---|---|")
for k in ("trunk_f16l_kernel<3>", "ub_layer_kernel", "xobj_rows_kernel<false>", "z64_kernel", "l2c_kernel<false>", "fps_table_kernel", "sa1_64_kernel", "m0_kernel"):
    a = agg.get(k)
    if not a:
        continue
    gui = a.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui) if gui else float("nan")
    mf = a.get("SQ_INSTS_MFMA", 0)
    other = (a.get("SQ_INSTS_VALU", 0) - mf) / mf if mf else float("nan")
    lds = a.get("SQ_INSTS_LDS", 0)
    vpl = a.get("SQ_INSTS_VALU", 0) / lds if lds else float("nan")
    wc = a.get("SQ_WAVE_CYCLES", 0)
    ldsb = (4 * lds + a.get("SQ_LDS_BANK_CONFLICT", 0)) / (gui * 256) if gui else float("nan")
    print(f"| `{k}` | {100 * busy:.0f}% | {other:.1f} | {vpl:.1f} | {100 * a.get('SQ_WAIT_ANY', 0) / wc:.0f}% | {100 * a.get('SQ_WAIT_INST_ANY', 0) / wc:.0f}% | "
          f"{100 * 4 * lds / (gui * 256):.0f}% + {100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / (gui * 256):.1f}% |")
print("\n## Raw\n```")
print("\n".join(raw))
print("```")
