#!/usr/bin/env python3
"""One steady-state train step of a rocprofv3 --kernel-trace CSV as a compact table (for offline reading of the launch sequence):
start / end relative to the step's first kernel (us), gap to the previous kernel of the same queue, queue, grid, LDS, registers, name.

    python scripts/step_dump.py <kernel_trace.csv> <out.tsv> [step index]
"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if "pack_all_kernel" in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(cuts) // 2
step = rows[cuts[k]:cuts[k + 1] + 1]
t0 = int(step[0]["Start_Timestamp"])
last_end = {}
with open(sys.argv[2], "w") as out:
    out.write("start_us\tend_us\tdur_us\tgap_us\tqueue\tgrid\twg\tlds\tvgpr\tagpr\tsgpr\tscratch\tname\n")
    for r in step:
        s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dc::", "")
        if name.startswith("_ZN2dc"):
            name = name[6:40]
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
        out.write(f"{(s - t0) / 1e3:.2f}\t{(e - t0) / 1e3:.2f}\t{(e - s) / 1e3:.2f}\t{gap:.2f}\t{q}\t{grid // max(wg, 1)}\t{wg}\t"
                  f"{r.get('LDS_Block_Size', '')}\t{r.get('VGPR_Count', '')}\t{r.get('Accum_VGPR_Count', '')}\t{r.get('SGPR_Count', '')}\t"
                  f"{r.get('Scratch_Size', r.get('Private_Segment_Size', ''))}\t{name[:60]}\n")
print(f"step {k}: {len(step) - 1} dispatches written to {sys.argv[2]}")
