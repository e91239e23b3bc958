#!/bin/bash
# usage: pmc.sh outname "COUNTERS..." -- program args   (one counter pass, csv under gpurun_out/)
out=$1; ctr=$2; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$out
timeout -s KILL 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$out -- "$@" > /tmp/pmc_$out.log 2>&1 || { tail -5 /tmp/pmc_$out.log; exit 1; }
f=$(find /tmp/pmc_$out -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,v in acc.items():
    if "vsm" not in k: continue
    print(k, "dispatches", len(n[k]))
    for c,x in sorted(v.items()): print("    %-28s %16.0f per dispatch" % (c, x/len(n[k])))
PY
