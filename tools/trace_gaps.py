#!/usr/bin/env python3
"""Idle time between consecutive conv-path kernels in a rocprofv3 kernel trace: tools/trace_gaps.py <trace_kernel_trace.csv>"""
import csv, sys
allrows = list(csv.DictReader(open(sys.argv[1])))
from collections import Counter
print("kernels in the trace:", dict(Counter(r["Kernel_Name"].split("(")[0][-50:] for r in allrows)))
rows = [r for r in allrows if "srcnn" in r["Kernel_Name"] and "probe" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                       # skip warm-up
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = sorted(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:]))
pairs = Counter()
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    pairs[(a["Kernel_Name"].split("(")[0][-28:], b["Kernel_Name"].split("(")[0][-28:], "q" + a["Queue_Id"] + "->q" + b["Queue_Id"])] += g
print({k: round(v / 1e3, 1) for k, v in pairs.items()})
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({busy / span:.3f}); gaps: median {gaps[len(gaps) // 2] / 1e3:.1f} us, "
      f"p90 {gaps[int(len(gaps) * .9)] / 1e3:.1f}, max {gaps[-1] / 1e3:.1f}, sum {sum(g for g in gaps if g > 0) / 1e6:.3f} ms")
