#!/bin/bash
# usage: run_variants.sh workload variant...   (timing experiments only: swaps the library in the box's scratch copy)
set -e
w=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp /root/repo/tools/exp/libvsm_$v.so /root/repo/visual_slam_amd/libvsm.so
  rm -rf /tmp/prof_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 /root/repo/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > /tmp/out_$v.log 2>&1
  echo "== $v: $(python3 -c "import json,sys; d=json.loads(open('/tmp/out_$v.log').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  f=$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("void vsm","vsm::")): print("   %-40s calls %5s avg %10.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
