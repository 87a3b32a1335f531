#!/usr/bin/env python3
"""Print every kernel (all streams) of the build phase of the last bench step in a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dgdm::", ""), r["Stream_Id"]) for r in rows)
tr = [e for e in ev if e[2].startswith("trunk")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
a, b = tr[k][1], tr[k + 1][0]
agg = {}
for s, e, n, q in ev:
    if s >= a and e <= b:
        if "copyBuffer" in n or "at::" in n: continue
        key = (n[:22], q)
        c = agg.setdefault(key, [0, 0.0, s, e]); c[0] += 1; c[1] += e - s; c[2] = min(c[2], s); c[3] = max(c[3], e)
print(f"window {(b-a)/1e6:.2f} ms")
for (n, q), (c, d, s, e) in sorted(agg.items(), key=lambda kv: kv[1][2]):
    print(f"  stream {q} {n:24s} x{c:3d} sum {d/1e6:7.3f} ms  first {(s-a)/1e6:7.3f} last_end {(e-a)/1e6:7.3f}")
