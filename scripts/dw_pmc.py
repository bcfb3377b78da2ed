"""Counter study of the depthwise kernels: why 728 channels at 48 x 72 move 3.4 TB/s where 128 channels at 384 x 576 move 5.5.
Workload (no arguments): forward on the tiled and on the pipelined kernel, fused data gradient on the pipelined one, each 6 x on rotating
buffers, on both shapes.  Run it under rocprofv3 once per counter set (scripts/dw_pmc.sh), then
    python scripts/dw_pmc.py reduce <dir with the passes> [out.txt]"""
import csv, glob, os, sys
from collections import OrderedDict, defaultdict


def workload():
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    from mlperf_deepcam_amd import lib as L
    dev = torch.device("cuda", 0); dt = L.DC_BF16; lib = L.load(); st = L.stream_ptr(); P = L.dptr
    NB, REPS = 3, 6
    for (Cc, H, W, N, dil) in [(728, 48, 72, 8, 1), (128, 384, 576, 8, 1), (256, 192, 288, 8, 1)]:
        ld = (Cc + 63) // 64 * 64
        act = lambda: [torch.randn(N, H, W, ld, device=dev).to(torch.bfloat16) for _ in range(NB)]
        x, y, dy, dx = act(), act(), act(), act()
        wp = torch.randn(9 * Cc, device=dev) * 0.2
        sc, sh, mean, invstd = [torch.rand(Cc, device=dev) + 0.5 for _ in range(4)]
        for mode in (0, 2):
            L.call("dc_set_option", b"dw_pipe", mode)
            for r in range(REPS):
                i = r % NB
                L.call("dc_dwconv_fwd", dt, Cc, 1, dil, N, H, W, P(x[i]), ld, P(wp), P(y[i]), ld, None, None, 0, st)
        L.call("dc_set_option", b"dw_pipe", 1)
        rows = lib.dc_dwconv_dgrad_bnstats_rows(dt, Cc, 1, dil, N, H, W); wrows = lib.dc_dwconv_dgrad_wgrad_rows(dt, Cc, 1, dil, N, H, W)
        slab = torch.empty(2 * rows * Cc, device=dev); wslab = torch.empty(max(wrows, 1) * 9 * Cc, device=dev)
        for r in range(REPS):
            i = r % NB
            L.call("dc_dwconv_dgrad", dt, Cc, 1, dil, N, H, W, P(dy[i]), ld, P(wp), None, 0, P(dx[i]), ld, st)
        if wrows > 0:
            for r in range(REPS):
                i = r % NB
                L.call("dc_dwconv_dgrad_bnstats_wgrad", dt, Cc, 1, dil, N, H, W, P(dy[i]), ld, P(wp), P(dx[i]), ld, P(y[i]), ld, P(mean), P(invstd),
                       P(sc), P(sh), 1, P(slab), P(wslab), st)
        # the yardstick on the same tensors: an elementwise pass (BatchNorm apply), one read and one write of M x C
        for r in range(REPS):
            i = r % NB
            L.call("dc_bn_apply", dt, N * H * W, Cc, P(x[i]), ld, P(sc), P(sh), None, 0, 1, P(y[i]), ld, st)
        torch.cuda.synchronize()


def reduce(root, out):
    disp = OrderedDict()
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        pas = os.path.relpath(path, root).split(os.sep)[0]
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if not any(k in name for k in ("dwp_kernel", "dwt_kernel", "bn_apply")):
                continue
            short = name.replace("void ", "").replace("dc::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
            key = (short, int(r["Grid_Size"]))
            d = disp.setdefault(key, {"n": defaultdict(int), "c": defaultdict(float), "t": defaultdict(float)})
            d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
            d["n"][r["Counter_Name"]] += 1
            d["t"][r["Counter_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    lines = []
    for (short, grid), d in disp.items():
        avg = {k: d["c"][k] / d["n"][k] for k in d["c"]}
        us = {k: d["t"][k] / d["n"][k] for k in d["c"]}
        lines.append(f"{short}  grid {grid}")
        for k in sorted(avg):
            lines.append(f"    {k:38s} {avg[k]:16.0f}   per us {avg[k] / us[k]:12.1f}   ({us[k]:.1f} us under this pass)")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "reduce":
        reduce(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        workload()
