#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace csv: per bench step, busy time of the union of kernels, idle gaps,
per-kernel exclusive time on the critical stream.  usage: trace_gaps.py <kernel_trace.csv> [t_from_frac]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dgdm::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
lo = t0 + (t1 - t0) * frac
ev = [e for e in ev if e[0] >= lo]
t0 = ev[0][0]
# union busy
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
for s, e, n, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print(f"span {span/1e6:.2f} ms, union busy {busy/1e6:.2f} ms, idle {100*(1-busy/span):.1f}%")
gaps.sort(reverse=True)
print("largest gaps (ms, at ms, next kernel):", [(round(g/1e6, 3), round(a/1e6, 1), n[:20]) for g, a, n in gaps[:12]])
print("total gap in gaps>20us: %.2f ms" % (sum(g for g, _, _ in gaps if g > 20000) / 1e6))
perq = defaultdict(float); perk = defaultdict(float)
for s, e, n, q in ev:
    perq[q] += e - s; perk[n] += e - s
print("per stream/queue busy ms:", {k: round(v/1e6, 1) for k, v in perq.items()})
print("per kernel ms:", {k[:24]: round(v/1e6, 1) for k, v in sorted(perk.items(), key=lambda kv: -kv[1])[:12]})
# time where ONLY non-trunk kernels run (trunk not active)
tr = [(s, e) for s, e, n, q in ev if n.startswith("trunk")]
tb = sum(e - s for s, e in tr)
print(f"trunk active {tb/1e6:.2f} ms = {100*tb/span:.1f}% of span")
