#!/usr/bin/env python3
"""Aggregate a scripts/step_dump.py table by kernel (and optionally by launch shape):  python scripts/serial_agg.py <tsv> [--shapes]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1]), delimiter="\t"))
shapes = "--shapes" in sys.argv
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["name"][:44] + (f" g{r['grid']}" if shapes else "")
    agg[n][0] += 1
    agg[n][1] += float(r["dur_us"])
tot = sum(v[1] for v in agg.values())
print(f"{len(rows)} launches, {tot / 1e3:.3f} ms of kernel time, span {float(rows[-1]['end_us']) / 1e3:.3f} ms")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:9.1f} us {v[0]:4d} x {v[1] / v[0]:7.1f}  {100 * v[1] / tot:5.1f} %  {n}")
