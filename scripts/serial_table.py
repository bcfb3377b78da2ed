#!/usr/bin/env python3
"""One steady-state step, every kernel ALONE on the GPU (DC_SIDE_STREAM=0) beside the same kernels in the two-stream step: two tables of
scripts/step_dump.py (collect_profiles.sh writes step_serial.tsv / step_both.tsv) reduced to one line per kernel.

    python scripts/serial_table.py <step_serial.tsv> <step_both.tsv>"""
import collections, csv, sys


def load(path):
    rows = list(csv.DictReader(open(path), delimiter="\t"))
    agg = collections.OrderedDict()
    for r in rows:
        n = r["name"][:58]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += float(r["dur_us"])
    span = float(rows[-1]["end_us"]) - float(rows[0]["start_us"])
    return rows, agg, span


srows, ser, sspan = load(sys.argv[1])
brows, both, bspan = load(sys.argv[2])
stot, btot = sum(v[1] for v in ser.values()), sum(v[1] for v in both.values())
print(f"serial: {len(srows)} dispatches, {stot / 1e3:.2f} ms of kernel time; two streams: {btot / 1e3:.2f} ms of kernel time in a {bspan / 1e3:.2f} ms step\n")
print(f"{'kernel':<60} {'launches':>8} {'alone ms':>9} {'avg us':>8} {'two-stream ms':>13}")
for n, v in sorted(ser.items(), key=lambda kv: -kv[1][1]):
    b = both.get(n, [0, 0.0])
    print(f"{n:<60} {v[0]:>8d} {v[1] / 1e3:>9.3f} {v[1] / v[0]:>8.1f} {b[1] / 1e3:>13.3f}")
