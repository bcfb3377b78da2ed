#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace CSV to: per-kernel totals, union-busy time, and idle gaps between kernels.

    python scripts/trace_gaps.py <kernel_trace.csv> [n_steps_to_skip_from_the_front]
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.1f} ms")
# union of busy intervals
busy = 0
cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
for s, e, n, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"union busy {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
small = [g for g, _ in gaps if g < 50_000]
print(f"gaps < 50 us: {len(small)}, sum {sum(small) / 1e6:.2f} ms, mean {sum(small) / max(1, len(small)) / 1e3:.2f} us")
byq = defaultdict(lambda: [0, 0])
for s, e, n, q in rows:
    byq[q][0] += 1
    byq[q][1] += e - s
for q, (c, t) in byq.items():
    print(f"queue {q}: {c} dispatches, {t / 1e6:.1f} ms kernel time")
after = defaultdict(lambda: [0, 0])
for g, n in gaps:
    if g < 50_000:
        k = n.split("(")[0][:60]
        after[k][0] += 1
        after[k][1] += g
print("idle before kernel (top 15):")
for k, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {t / 1e3:9.1f} us  x{c:5d}  {t / c / 1e3:6.2f} us each  {k}")
