"""Start/end times of the last N kernel dispatches of a rocprofv3 --kernel-trace run, relative to the first of them
(what a single call queues, and the gaps between its kernels).   usage: python tools/kernel_timeline.py TRACE.csv [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void vsm::", "").replace("vsm::", "")
    gap = "" if prev_end is None else "  gap %6.1f us" % ((s - prev_end) / 1e3)
    print("%8.1f .. %8.1f us  (%6.1f us)  %s%s" % (s / 1e3, e / 1e3, (e - s) / 1e3, name[:60], gap))
    prev_end = e
