#!/usr/bin/env python3
"""One steady-state train step out of a rocprofv3 --kernel-trace CSV: how the two HIP queues (backward chain / weight gradients)
share the GPU.  Steps are cut at dc::pack_all_kernel (first kernel of a step's forward).

    python scripts/step_timeline.py <kernel_trace.csv> [step index, default: the middle one]
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
rows.sort()
cuts = [i for i, r in enumerate(rows) if "pack_all_kernel" in r[2]]
if len(cuts) < 3:
    sys.exit("need at least three steps in the trace")
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(cuts) // 2
step = rows[cuts[k]:cuts[k + 1]]
t0, t1 = step[0][0], rows[cuts[k + 1]][0]
print(f"step {k}: {len(step)} dispatches, {(t1 - t0) / 1e6:.2f} ms from pack_all to the next pack_all")
queues = sorted({r[3] for r in step}, key=lambda q: -sum(1 for r in step if r[3] == q))
main = queues[0]


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


um = union([(s, e) for s, e, n, q in step if q == main])
us = union([(s, e) for s, e, n, q in step if q != main])
both = intersect(um, us)
print(f"main queue busy {length(um) / 1e6:.2f} ms, other queue(s) busy {length(us) / 1e6:.2f} ms, both at once {length(both) / 1e6:.2f} ms, "
      f"neither {((t1 - t0) - length(um) - length(us) + length(both)) / 1e6:.2f} ms")
last_main = max(e for s, e, n, q in step if q == main)
last_side = max([e for s, e, n, q in step if q != main] or [t0])
first_side = min([s for s, e, n, q in step if q != main] or [t0])
print(f"side queue active from {(first_side - t0) / 1e6:.2f} to {(last_side - t0) / 1e6:.2f} ms; main queue's last kernel ends at {(last_main - t0) / 1e6:.2f} ms")
# kernel time per queue by kernel
for q in queues:
    tot = defaultdict(lambda: [0, 0])
    for s, e, n, qq in step:
        if qq == q:
            tot[n][0] += 1
            tot[n][1] += e - s
    print(f"queue {q}: {sum(v[0] for v in tot.values())} dispatches, {sum(v[1] for v in tot.values()) / 1e6:.2f} ms of kernel time")
    for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {t / 1e6:7.3f} ms  x{c:4d}  {n[:90]}")
# 2 ms bins: share of time with main busy / side busy
print("bins of 2 ms: main-busy / side-busy fraction")
b = t0
while b < t1:
    e = min(b + 2_000_000, t1)
    w = [[b, e]]
    print(f"   {(b - t0) / 1e6:5.1f} ms  main {length(intersect(um, w)) / (e - b):4.2f}  side {length(intersect(us, w)) / (e - b):4.2f}")
    b = e
