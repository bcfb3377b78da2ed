"""One steady-state step out of a rocprofv3 kernel trace, kernel by kernel (queue, start, duration, workgroups, name): the long poles of the chain.
    python scripts/step_kernels.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if "pack_all" in r["Kernel_Name"]]
a, b = packs[-2], packs[-1]            # the last complete step
t0 = int(rows[a]["Start_Timestamp"])
queues = {}
for r in rows[a:b]:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if dur >= min_us:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dc::", "").replace("(anonymous namespace)::", "")[:60]
        print(f"q{q} {(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} ms {dur:8.1f} us  wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d}  {name}")
