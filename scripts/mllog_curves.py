#!/usr/bin/env python3
"""Loss / IoU curves out of the driver's ``:::MLLOG`` files: the text analogue of the reference's analysis/training_analysis.ipynb
(cells at :84-200 parse the MLPerf log lines of one or several runs into per-step train_loss / train_accuracy / eval_accuracy
series and plot them against each other).

    python scripts/mllog_curves.py runA.log [runB.log ...] [--target 0.82] [--every 24] [--csv out.csv]

For every run: steps logged, first step at which eval_accuracy reaches the target (the driver's stop rule, train_hdf5_ddp.py:505-507),
final train loss / IoU.  With two or more runs the FIRST one is the baseline (e.g. the fp32 engine): per logging interval the
relative gap of the mean train loss, its maximum, and the eval curves side by side -- the honest form of "loss curve within x of
the reference" for runs whose trajectories separate after the first update.
"""
import argparse
import json
import os
import sys
from collections import OrderedDict


def parse(path):
    """-> {"meta": {key: value}, "series": {key: [(step, value), ...]}} of one :::MLLOG file (or a stdout capture that contains
    such lines)."""
    meta, series = {}, {}
    with open(path) as f:
        for line in f:
            i = line.find(":::MLLOG ")
            if i < 0:
                continue
            try:
                e = json.loads(line[i + len(":::MLLOG "):])
            except json.JSONDecodeError:
                continue
            key, val, md = e.get("key"), e.get("value"), e.get("metadata") or {}
            if "step_num" in md and isinstance(val, (int, float)):
                series.setdefault(key, []).append((int(md["step_num"]), float(val)))
            elif key not in meta:
                meta[key] = val
    return {"meta": meta, "series": series}


def first_reaching(points, target):
    for step, v in points:
        if v >= target:
            return step, v
    return None


def interval_means(points, every):
    out = OrderedDict()
    for step, v in points:
        out.setdefault((step - 1) // every, []).append(v)
    return OrderedDict((k, sum(v) / len(v)) for k, v in out.items())


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("logs", nargs="+")
    ap.add_argument("--target", type=float, default=0.82)
    ap.add_argument("--every", type=int, default=24, help="steps per comparison interval")
    ap.add_argument("--csv", default=None, help="write step,run,key,value rows")
    a = ap.parse_args()
    runs = [(os.path.basename(p), parse(p)) for p in a.logs]
    rows = []
    for name, r in runs:
        s = r["series"]
        tl, ta, ea = s.get("train_loss", []), s.get("train_accuracy", []), s.get("eval_accuracy", [])
        hit = first_reaching(ea, a.target)
        print(f"== {name}: global_batch {r['meta'].get('global_batch_size')}, optimizer {r['meta'].get('opt_name')}, "
              f"{len(tl)} train points, {len(ea)} evaluations")
        if tl:
            print(f"   train_loss  first {tl[0][1]:.6f} (step {tl[0][0]})  last {tl[-1][1]:.6f} (step {tl[-1][0]})  min {min(v for _, v in tl):.6f}")
        if ta:
            print(f"   train IoU   first {ta[0][1]:.4f}  last {ta[-1][1]:.4f}  max {max(v for _, v in ta):.4f}")
        if ea:
            print("   eval IoU    " + "  ".join(f"{st}:{v:.3f}" for st, v in ea))
        print(f"   steps to eval IoU >= {a.target}: " + (f"{hit[0]} (IoU {hit[1]:.4f})" if hit else "not reached"))
        for key in ("train_loss", "train_accuracy", "eval_accuracy", "eval_loss", "learning_rate"):
            rows += [(st, name, key, v) for st, v in s.get(key, [])]
    if len(runs) > 1:
        base_name, base = runs[0]
        bm = interval_means(base["series"].get("train_loss", []), a.every)
        for name, r in runs[1:]:
            om = interval_means(r["series"].get("train_loss", []), a.every)
            common = [k for k in bm if k in om]
            print(f"== train_loss of {name} against {base_name}, means over {a.every}-step intervals")
            worst = 0.0
            for k in common:
                gap = (om[k] - bm[k]) / abs(bm[k]) if bm[k] else 0.0
                worst = max(worst, abs(gap))
                print(f"   steps {k * a.every + 1:5d}-{(k + 1) * a.every:5d}: {bm[k]:.6f} vs {om[k]:.6f}   rel gap {gap:+.3e}")
            b0, o0 = base["series"].get("train_loss", [(0, 0.0)])[0], r["series"].get("train_loss", [(0, 0.0)])[0]
            if b0[0] == o0[0] and b0[1]:
                print(f"   step {b0[0]} (before any update can differ): rel gap {(o0[1] - b0[1]) / abs(b0[1]):+.3e}")
            print(f"   max |relative gap| of the interval means: {worst:.3e}")
            be, oe = dict(base["series"].get("eval_accuracy", [])), dict(r["series"].get("eval_accuracy", []))
            both = sorted(set(be) & set(oe))
            if both:
                print("   eval IoU    " + "  ".join(f"{st}:{be[st]:.3f}/{oe[st]:.3f}" for st in both))
            hb, ho = first_reaching(base["series"].get("eval_accuracy", []), a.target), first_reaching(r["series"].get("eval_accuracy", []), a.target)
            print(f"   steps to target: {hb[0] if hb else None} vs {ho[0] if ho else None}")
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("step,run,key,value\n")
            for st, name, key, v in sorted(rows):
                f.write(f"{st},{name},{key},{v}\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
