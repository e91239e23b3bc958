#!/bin/bash
# The bench line of every workload without a profiler attached (run on the GPU box; lines land in gpurun_out/prof/).
# usage: tools/refresh_unprofiled.sh TAG
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
for w in db_query superpoint_pair superpoint_pairs sift_large_pair; do
  extra=""
  [ $w = superpoint_pair ] && extra="--steps 200"
  timeout -k 10 300 python3 $root/bench.py --workload $w $extra > $out/${tag}_bench_${w}.json.log 2>/dev/null || exit 1
  echo "$w: $(python3 -c "import json; d=json.loads(open('$out/${tag}_bench_${w}.json.log').read().strip().splitlines()[-1]); print('%.4f ms/step verified=%s' % (d['ms_per_step'], d['verified']))")"
done
