"""HBM-side traffic per launch of the MFMA kernel families from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950).  Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE is tallied at 64 B per 128-B request -> doubled; WRITE_SIZE
as read; both counters are in KiB.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/fetch -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/write -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline
    python scripts/pmc_traffic.py out/fetch out/write profiles/r01_pmc_traffic.json
"""
import csv, glob, json, os, sys

import re

FAMILIES = {"igemm": ("igemm_kernel", "igemm256_kernel", "igemm256p_kernel", "pw384_kernel", "pw224_kernel", "pw192_kernel", "tiny_gemm_kernel"),
            "wgrad": ("wgrad_dma_kernel", "wgrad256_kernel", "wgrad384_kernel", "wgrad_kernel", "wgrad_reduce_kernel", "fold_kernel")}
# Every dispatched kernel that LOOKS like a member of one of the two families must be listed above: round 3's "1.05 x" for the implicit GEMMs
# came from a table that silently lacked igemm256p_kernel.  Depthwise / thin-layer / slab-statistics kernels are other families.
FAMILY_LIKE = re.compile(r"(igemm\w*_kernel|pw\d+_kernel|tiny_gemm_kernel|(?<![a-z_])wgrad\w*_kernel|(?<![a-z_])fold_kernel)")
NOT_FAMILY = ("dw_wgrad", "dwt_wgrad", "dws2_wgrad", "thin_wgrad", "slab_fold", "head_")
# launches of the C-ABI entry point = launches of the main kernel (the fold of the slabs rides along with each weight-gradient launch; since
# round 4 it also folds the depthwise layers' rows, a few MB per launch)
MAIN = {"igemm": ("igemm_kernel", "igemm256_kernel", "igemm256p_kernel", "pw384_kernel", "pw224_kernel", "pw192_kernel", "tiny_gemm_kernel"),
        "wgrad": ("wgrad_dma_kernel", "wgrad256_kernel", "wgrad384_kernel", "wgrad_kernel")}


def collect(d, counter):
    tot = {f: 0.0 for f in FAMILIES}
    calls = {f: 0 for f in FAMILIES}
    per_kernel = {}
    unlisted = set()
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            m = FAMILY_LIKE.search(name)
            if m and not any(x in name for x in NOT_FAMILY) and not any(k in name for keys in FAMILIES.values() for k in keys):
                unlisted.add(name)
            for fam, keys in FAMILIES.items():
                hit = next((k for k in keys if k in name and "dw" not in name.split(k)[0][-4:]), None)
                if hit and "dw_wgrad" not in name and "dwt_wgrad" not in name:
                    tot[fam] += float(r["Counter_Value"])
                    pk = per_kernel.setdefault(hit, [0, 0.0])
                    pk[0] += 1
                    pk[1] += float(r["Counter_Value"])
                    if hit in MAIN[fam]:
                        calls[fam] += 1
    if unlisted:
        sys.exit("pmc_traffic.py: kernels of the GEMM families in the trace that the family tables do not list (add them): " + ", ".join(sorted(unlisted)))
    return tot, calls, per_kernel


fetch, calls_f, pk_f = collect(sys.argv[1], "FETCH_SIZE")
write, calls_w, pk_w = collect(sys.argv[2], "WRITE_SIZE")
cfg = {"local_batch": int(sys.argv[4]) if len(sys.argv) > 4 else 8, "dtype": "bf16", "height": 768, "width": 1152}
out = {"config": cfg, "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline",
       "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as read; KiB -> bytes; "
                     "per launch = per C-ABI call of the family (dc_conv_fwd + dc_conv_dgrad / dc_conv_wgrad incl. its slab reduction)",
       "kernels": {}, "per_kernel": {}}
for fam in FAMILIES:
    n = max(calls_f[fam], 1)
    out["kernels"][fam] = {"dispatches": calls_f[fam], "fetch_size_kib_avg": fetch[fam] / n, "write_size_kib_avg": write[fam] / max(calls_w[fam], 1),
                           "hbm_bytes_per_launch": (2.0 * fetch[fam] / n + write[fam] / max(calls_w[fam], 1)) * 1024.0}
for k in sorted(set(pk_f) | set(pk_w)):
    nf, vf = pk_f.get(k, [0, 0.0])
    nw, vw = pk_w.get(k, [0, 0.0])
    out["per_kernel"][k] = {"dispatches": nf, "hbm_bytes_per_dispatch": (2.0 * vf / max(nf, 1) + vw / max(nw, 1)) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
