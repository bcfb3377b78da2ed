"""Where the wave-cycles of each kernel go, from one rocprofv3 PMC pass over the bench (SQ block, 8 counters):

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
              SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -d out/sq -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline
    python scripts/pmc_sq.py out/sq [out.json]

WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md); all in
quad-cycles, summed over the kernel's dispatches."""
import csv, glob, json, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(float)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        key = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[key] += 1; dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for k, c in acc.items():
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if w <= 0: continue
    pct = lambda name: round(100.0 * c.get(name, 0.0) / w, 1)
    rows.append({"kernel": k, "dispatches": n[k], "total_us": round(dur[k], 1), "wait_any": pct("SQ_WAIT_ANY"), "wait_inst": pct("SQ_WAIT_INST_ANY"),
                 "active_any": pct("SQ_ACTIVE_INST_ANY"), "valu": pct("SQ_ACTIVE_INST_VALU"), "lds": pct("SQ_ACTIVE_INST_LDS"), "vmem": pct("SQ_ACTIVE_INST_VMEM"),
                 "lds_conflict_vs_wave": pct("SQ_LDS_BANK_CONFLICT")})
rows.sort(key=lambda r: -r["total_us"])
if len(sys.argv) > 2: json.dump({"kernels": rows}, open(sys.argv[2], "w"), indent=1)
print(f"{'us':>9} {'n':>5} {'wait':>6} {'stall':>6} {'active':>6} {'valu':>6} {'lds':>6} {'vmem':>6} {'ldsconf':>7}  kernel")
for r in rows[:28]:
    print(f"{r['total_us']:9.0f} {r['dispatches']:5d} {r['wait_any']:6.1f} {r['wait_inst']:6.1f} {r['active_any']:6.1f} {r['valu']:6.1f} {r['lds']:6.1f} {r['vmem']:6.1f} {r['lds_conflict_vs_wave']:7.1f}  {r['kernel'][:70]}")
