"""Per-kernel averages of the library's own kernels out of a rocprofv3 --kernel-trace --stats summary
(the torch kernels that build the synthetic data are left out).   usage: python tools/vsm_kernel_times.py STATS.csv"""
import csv
import sys

for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"]
    if "vsm::" not in name:
        continue
    short = name.split("vsm::", 1)[1].replace("(anonymous namespace)::", "").split("(", 1)[0]
    print("%-44s calls %5s  avg %9.3f us  total %9.3f ms" % (short, row["Calls"], float(row["AverageNs"]) / 1e3,
                                                            float(row["TotalDurationNs"]) / 1e6))
