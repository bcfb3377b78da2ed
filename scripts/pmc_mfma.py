"""MFMA utilisation per kernel from one rocprofv3 PMC pass (separate from the kernel-trace/stats run and from the FETCH/WRITE passes):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d out/mfma -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline
    python scripts/pmc_mfma.py out/mfma profiles/r01_pmc_mfma.json

util = sum over SIMDs of SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 1024 SIMDs): the fraction of SIMD-cycles with the matrix pipe busy
while the kernel had the GPU (the expression of rocprofv3's MfmaUtil, taken from the raw counters).  GRBM_GUI_ACTIVE is reported as the SUM
over the 8 XCDs (checked against the dispatch's own timestamps: 106.8 us x 2.1 GHz x 8 = 1.79 M for a value of 1.87 M), so active cycles =
GRBM_GUI_ACTIVE / 8.  Kernels run one at a time under counter collection, so these are standalone figures.  A bf16 16x16x32 MFMA keeps the
pipe busy 16 cycles (4 passes), so 100 % here = the 2.5 PFLOP/s dense peak at the clock the kernel actually ran at."""
import csv, glob, json, os, sys
from collections import defaultdict

SIMDS = 256 * 4
XCDS = 8
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[key] += 1
rows = []
for k, c in acc.items():
    busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if busy > 0 and act > 0:
        rows.append({"kernel": k[:90], "dispatches": n[k], "gpu_active_cycles": act, "mfma_busy_simd_cycles": busy,
                     "mfma_util_percent": round(100.0 * busy / (act / XCDS * SIMDS), 2)})
rows.sort(key=lambda r: -r["gpu_active_cycles"])
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline",
       "definition": "100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), per kernel over all its dispatches", "kernels": rows}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for r in rows:
    print(f"{r['mfma_util_percent']:6.2f} %  x{r['dispatches']:5d}  {r['kernel']}")
