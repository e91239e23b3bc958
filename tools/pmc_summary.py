"""Folds rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel.

usage: pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json

Both counters are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide
streaming reads (16 B per lane, global loads and LDS-DMA alike) at 64 bytes, i.e. reads half the
bytes: it is doubled here.  WRITE_SIZE is taken as read.
"""
import collections
import csv
import json
import sys


def fold(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, nf = fold(sys.argv[1], "FETCH_SIZE")
    write, _ = fold(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * 1024.0 * fetch.get(k, 0.0)
        wr = 1024.0 * write.get(k, 0.0)
        out[k] = {"launches": nf.get(k, 0), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    json.dump({"unit": "bytes", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950), WRITE_SIZE KiB x 1024",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print("%-40s %12.0f B read %12.0f B written per launch" % (k[:40], v["read_bytes_per_launch"], v["write_bytes_per_launch"]))


if __name__ == "__main__":
    main()
