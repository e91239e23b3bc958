#!/bin/bash
# Regenerates only the three files of the default workload (see tools/refresh_profiles.sh for the whole set).
set -e
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $root/bench.py > $out/${tag}_bench_default.json.log 2>/tmp/bench_default.err
rm -rf /tmp/prof_default
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -- python3 $root/bench.py --no-cpu-baseline > $out/${tag}_bench_default_under_rocprof.json.log 2>/tmp/prof_default.err
cp "$(find /tmp/prof_default -name '*kernel_stats.csv' | head -1)" $out/${tag}_bench_default_kernel_stats.csv
echo done
