"""Prints the top rows of a rocprofv3 kernel_stats.csv: python scripts/prof_top.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
