#!/usr/bin/env python3
"""Per-kernel roofline table of the train step from three rocprofv3 PMC passes over `bench.py --steps 2 --warmup 1 --no_cpu_baseline`
(the MI355X analogue of the reference's Nsight metric sweep, analysis/*.ipynb; ceilings from MI355X_MICROARCH.md instead of the V100
constants of analysis/roofline_plot.ipynb:84-92):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE                 -d out/FETCH_SIZE -o run --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE                 -d out/WRITE_SIZE -o run --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
                                                              -d out/SQ_INSTS_VALU_MFMA_MOPS_BF16 -o run --output-format csv -- python3 bench.py ...
    python scripts/roofline_report.py out profiles/r01_roofline_table.md

Per kernel (all its dispatches of the profiled steps, kernels serialised by the counter collection):
  bytes   = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH doubled on gfx950 as the guide prescribes): fabric-side traffic of the L2s,
            i.e. HBM + Infinity-Cache, an upper bound of the HBM bytes
  flop    = 512 x (MOPS_BF16 + MOPS_F32): what the matrix pipes executed (padding included), not the algorithmic count
  time    = dispatch end - start from the same pass as the MOPS counters
  bound   = min(2.5 PFLOP/s, intensity x 8 TB/s); frac = achieved / bound
"""
import csv, glob, os, sys
from collections import defaultdict

PEAK_F, PEAK_B = 2.5e15, 8.0e12


def key_of(name):
    k = name.replace("void ", "").replace("(anonymous namespace)::", "")
    k = k.split("(")[0]
    if k.startswith("_ZN2dc"):                   # mangled: keep the function name and the first template argument's dtype
        import re
        m = re.match(r"_ZN2dc(?:12_GLOBAL__N_1)?(\d+)", k)
        if m:
            n = int(m.group(1)); start = m.end()
            k = "dc::" + k[start:start + n] + ("<bf16>" if "DF16b" in k else "<f32>" if "IfL" in k or "IfE" in k else "")
    return k[:64]


SHAPES = defaultdict(lambda: defaultdict(float))      # (kernel, workgroups) -> counter sums, time, dispatches: a launch shape = a layer geometry


def read(d, counters):
    acc = defaultdict(lambda: defaultdict(float)); t = defaultdict(float); n = defaultdict(int); seen = set()
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] not in counters:
                continue
            k = key_of(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if any(f in k for f in ("igemm", "pw384", "pw224", "pw192", "wgrad384", "wgrad_dma")):
                sk = (k, int(r["Grid_Size"]) // max(int(r.get("Workgroup_Size", 0) or 0), 1))
                SHAPES[sk][r["Counter_Name"]] += float(r["Counter_Value"])
                if (r["Dispatch_Id"], r["Counter_Name"]) not in seen and r["Counter_Name"] in ("FETCH_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
                    seen.add((r["Dispatch_Id"], r["Counter_Name"]))
                    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                        SHAPES[sk]["n"] += 1
                        SHAPES[sk]["t"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); n[k] += 1
                t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return acc, t, n


root = sys.argv[1]
f, _, _ = read(os.path.join(root, "FETCH_SIZE"), {"FETCH_SIZE"})
w, _, _ = read(os.path.join(root, "WRITE_SIZE"), {"WRITE_SIZE"})
m, t, n = read(os.path.join(root, "SQ_INSTS_VALU_MFMA_MOPS_BF16"),
               {"SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
rows = []
for k in t:
    if not k.startswith("dc::"):
        continue
    nbytes = (2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0
    flop = 512.0 * (m[k]["SQ_INSTS_VALU_MFMA_MOPS_BF16"] + m[k]["SQ_INSTS_VALU_MFMA_MOPS_F32"])
    busy, act = m[k]["SQ_VALU_MFMA_BUSY_CYCLES"], m[k]["GRBM_GUI_ACTIVE"] / 8.0
    secs = t[k]
    ai = flop / nbytes if nbytes > 0 else 0.0
    bound_f = min(PEAK_F, ai * PEAK_B) if flop > 0 else 0.0
    rows.append(dict(k=k, n=n[k], ms=secs * 1e3, gb=nbytes / 1e9, gbs=nbytes / secs / 1e9, tf=flop / secs / 1e12, ai=ai,
                     frac_b=nbytes / secs / PEAK_B, frac_f=(flop / secs / bound_f) if bound_f > 0 else 0.0,
                     util=100.0 * busy / (act * 1024.0) if act > 0 else 0.0, bound="MFMA" if ai * PEAK_B >= PEAK_F else "HBM"))
rows.sort(key=lambda r: -r["ms"])
steps = 3     # bench.py --steps 2 --warmup 1, plus its 9 roofline-pass steps: the table is per profiled run, shares are what matter
tot = sum(r["ms"] for r in rows)
out = ["| kernel | dispatches | time share | GB moved | GB/s | % of 8 TB/s | MFMA TFLOP/s | flop/byte | roof | % of its roof | MFMA util |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    if r["ms"] < 0.002 * tot:
        continue
    roof = r["bound"] if r["tf"] > 0 else "HBM"
    pct = 100.0 * (r["frac_f"] if (r["tf"] > 0 and roof == "MFMA") else r["frac_b"]) if roof == "HBM" or r["tf"] > 0 else 0.0
    if roof == "HBM":
        pct = 100.0 * r["frac_b"]
    out.append(f"| `{r['k']}` | {r['n']} | {100 * r['ms'] / tot:.1f} % | {r['gb']:.1f} | {r['gbs']:.0f} | {100 * r['frac_b']:.0f} % | "
               f"{r['tf']:.0f} | {r['ai']:.0f} | {roof} | {pct:.0f} % | {r['util']:.0f} % |")
steps = max(n.get(k, 0) for k in n if "pack_all" in k) if any("pack_all" in k for k in n) else 1
tot_gb = sum(r["gb"] for r in rows)
out.append("")
LB = int(sys.argv[3]) if len(sys.argv) > 3 else 8
out.append(f"**Fabric-side traffic of the whole step: {tot_gb / steps:.1f} GB per step = {tot_gb / steps / LB:.2f} GB per sample** "
           f"({steps} profiled steps of local batch {LB}; MFMA kernels {sum(r['gb'] for r in rows if r['tf'] > 0) / steps:.1f} GB, the others "
           f"{sum(r['gb'] for r in rows if r['tf'] <= 0) / steps:.1f} GB).  At the 6.29 TB/s a copy reaches that alone is "
           f"{tot_gb / steps / 6.29:.1f} ms per step.")
out.append("")
out.append("Implicit-GEMM and weight-gradient launches by launch shape (kernel x workgroups = one layer geometry), largest fabric traffic first:")
out.append("")
out.append("| kernel | workgroups | launches per step | us per launch | MB per launch (fabric side) | MFMA util |")
out.append("|---|---|---|---|---|---|")
shp = []
for (k, wgs), v in SHAPES.items():
    if v["n"] <= 0:
        continue
    mb = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / 1e6 / v["n"]
    act = v["GRBM_GUI_ACTIVE"] / 8.0
    shp.append((mb * v["n"], k, wgs, v["n"] / steps, v["t"] / v["n"] * 1e6, mb, 100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024.0) if act > 0 else 0.0))
for _, k, wgs, per, us, mb, util in sorted(shp, reverse=True)[:28]:
    out.append(f"| `{k}` | {wgs} | {per:.1f} | {us:.1f} | {mb:.1f} | {util:.0f} % |")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        fh.write("# Per-kernel roofline table\n\n" + __doc__.split("Per kernel")[0].strip().split("\n\n")[0] + "\n\n"
                 "Columns: bytes = 2 x FETCH_SIZE + WRITE_SIZE (fabric side of the L2s: HBM + Infinity Cache); flop = 512 x MFMA MOPS counters "
                 "(executed, padding included); roof = MFMA when flop/byte x 8 TB/s exceeds 2.5 PFLOP/s, else HBM; kernels run one at a time under "
                 "counter collection.  Produced by `scripts/roofline_report.py` (commands in its header).\n\n" + text + "\n")
