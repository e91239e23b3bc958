"""Folds a rocprofv3 --pmc counter_collection.csv into per-kernel, per-launch sums (one line per kernel)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
pick = sys.argv[2:] or None
for k, v in sorted(acc.items()):
    if pick and not any(p in k for p in pick):
        continue
    d = len(n[k])
    print(k, "launches", d, " ".join("%s=%.4g" % (c, x / d) for c, x in sorted(v.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        busy, act = v["SQ_VALU_MFMA_BUSY_CYCLES"] / d, v["GRBM_GUI_ACTIVE"] / d / 8.0
        print("   matrix-pipe busy / (elapsed cycles x 1024 SIMDs) = %.3f; elapsed cycles %.4g" % (busy / (act * 1024.0), act))
