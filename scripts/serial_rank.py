"""Per-kernel ranking of a `rocprofv3 --kernel-trace --stats` run of scripts/run_steps.py (DC_SIDE_STREAM=0: every kernel alone):
    python scripts/serial_rank.py <dir with *kernel_stats.csv> <steps incl. warm-up> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / n / 1e6:.2f} ms in {sum(int(r['Calls']) for r in rows) / n:.0f} dispatches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print(f"{float(r['TotalDurationNs']) / n / 1e6:7.3f} ms  {int(r['Calls']) / n:6.1f} x {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:90]}")
