import csv, glob, sys, os
from collections import OrderedDict
g = OrderedDict()
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "FETCH_SIZE" or "igemm256" not in r["Kernel_Name"]:
            continue
        k = int(r["Grid_Size"]) // 512
        v = g.setdefault(k, [0, 0.0])
        v[0] += 1; v[1] += float(r["Counter_Value"])
for k, (n, kb) in g.items():
    print(f"igemm256 x{k:5d} workgroups: {n:3d} launches, fetch {2 * kb * 1024 / n / 1e6:8.1f} MB per launch")
