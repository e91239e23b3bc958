#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box; results land in gpurun_out/prof/).
# usage: tools/refresh_profiles.sh TAG      e.g. r01
set -e
tag=${1:-r01}
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
timeout -k 10 300 python3 $root/bench.py --workload geom_verify > $out/${tag}_bench_geom_verify.json.log 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $root/tools/traffic_probe.py --images 1000 > /tmp/pmc_$c.log 2>&1
  cp "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" $out/${tag}_probe1000_pmc_$c.csv
  echo "pmc $c done"
done
python3 $root/tools/pmc_summary.py $out/${tag}_probe1000_pmc_FETCH_SIZE.csv $out/${tag}_probe1000_pmc_WRITE_SIZE.csv $out/${tag}_traffic_sift_exhaustive.json
