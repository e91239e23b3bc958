#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box; results land in gpurun_out/prof/).
# usage: tools/refresh_profiles.sh TAG      e.g. r03
set -e
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {   # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $root/bench.py "$@" --no-cpu-baseline > $out/${tag}_bench_${name}_under_rocprof.json.log 2>/tmp/prof_$name.err
  cp "$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)" $out/${tag}_bench_${name}_kernel_stats.csv
  echo "stats $name done"
}
timeout -k 10 400 python3 $root/bench.py > $out/${tag}_bench_default.json.log 2>/tmp/bench_default.err
echo "bench default done"
stats default
stats db_query --workload db_query
stats superpoint_pair --workload superpoint_pair --steps 50
stats superpoint_pairs --workload superpoint_pairs
stats sift_large_pair --workload sift_large_pair
stats geom_verify --workload geom_verify
stats shortlist --workload shortlist
timeout -k 10 300 python3 $root/bench.py --workload shortlist > $out/${tag}_bench_shortlist.json.log 2>/dev/null
timeout -k 10 300 python3 $root/bench.py --workload geom_verify > $out/${tag}_bench_geom_verify.json.log 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $root/tools/traffic_probe.py --images 1000 > /tmp/pmc_$c.log 2>&1
  cp "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" $out/${tag}_probe1000_pmc_$c.csv
  echo "pmc $c done"
done
python3 $root/tools/pmc_summary.py $out/${tag}_probe1000_pmc_FETCH_SIZE.csv $out/${tag}_probe1000_pmc_WRITE_SIZE.csv $out/${tag}_traffic_sift_exhaustive.json
# the same two passes over the float database query (16384 x 1 M x 256)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_db_$c
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_db_$c -- python3 $root/tools/db_query_probe.py > /tmp/pmc_db_$c.log 2>&1
  cp "$(find /tmp/pmc_db_$c -name '*counter_collection.csv' | head -1)" $out/${tag}_dbquery_pmc_$c.csv
done
python3 $root/tools/pmc_summary.py $out/${tag}_dbquery_pmc_FETCH_SIZE.csv $out/${tag}_dbquery_pmc_WRITE_SIZE.csv $out/${tag}_traffic_db_query.json
# matrix-core utilisation of the SIFT kernels: busy cycles of the matrix pipes against elapsed cycles x SIMDs
rm -rf /tmp/pmc_mfma
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python3 $root/tools/traffic_probe.py --images 300 > /tmp/pmc_mfma.log 2>&1
cp "$(find /tmp/pmc_mfma -name '*counter_collection.csv' | head -1)" $out/${tag}_probe300_pmc_mfma.csv
python3 - $out/${tag}_probe300_pmc_mfma.csv <<'PY' | tee $out/${tag}_mfma_utilisation.txt
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    if "sift_query" not in k and "sift_verify_kernel" not in k:
        continue
    d = len(n[k])
    # GRBM_GUI_ACTIVE comes summed over the 8 XCDs; the busy cycles are summed over all 1024 SIMDs
    busy, act = v["SQ_VALU_MFMA_BUSY_CYCLES"] / d, v["GRBM_GUI_ACTIVE"] / d / 8.0
    print("%s: %d launches; per launch: matrix-pipe busy cycles %.4g, elapsed cycles %.4g, busy / (elapsed x 1024 SIMDs) = %.3f; "
          "MFMA mops %.4g, VALU instructions %.4g, co-execution cycles %.4g"
          % (k, d, busy, act, busy / (act * 1024.0), v["SQ_INSTS_VALU_MFMA_MOPS_I8"] / d, v["SQ_INSTS_VALU"] / d,
             v["SQ_VALU_MFMA_COEXEC_CYCLES"] / d))
PY
