#!/bin/bash
# The default workload on every kind of synthetic descriptors bench.py can make, in the default form of the query
# pass and in the packed form, a few steps each (run on the GPU box; lines land in gpurun_out/prof/).
# Mode 0 adapts (a submission most of whose rows needed the rescan pass sends the next 31 to the packed form), mode 1 is
# the packed form, mode 3 the tile-minimum form whatever the data: its line on `duplicates` and `ones` is that form's
# worst case, and the kernel summary of the same command shows where the time goes.
# usage: tools/data_modes.sh TAG [STEPS]
tag=${1:-r03}
steps=${2:-6}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
for data in planted clustered duplicates ones; do
  for mode in 0 1 3; do
    timeout -k 10 240 python3 $root/bench.py --data $data --u8-mode $mode --steps $steps --warmup 2 --no-cpu-baseline \
      > $out/${tag}_bench_data_${data}_mode${mode}.json.log 2>$out/${tag}_bench_data_${data}_mode${mode}.err || exit 1
    echo "$data mode $mode: $(python3 -c "import json,sys; d=json.loads(open('$out/${tag}_bench_data_${data}_mode${mode}.json.log').read().strip().splitlines()[-1]); print('%.2f ms/step verified=%s kernel %.2f verify %.2f path %s matches/pair %.1f' % (d['ms_per_step'], d['verified'], d['roofline']['kernel_ms'], d['roofline']['verify_kernel_ms'], d['roofline']['kernel'], d['config']['matches_per_pair']))")"
  done
  rm -rf /tmp/prof_data_$data
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_data_$data -- python3 $root/bench.py --data $data --u8-mode 3 --steps $steps --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>/tmp/prof_data_$data.err) || exit 1
  cp "$(find /tmp/prof_data_$data -name '*kernel_stats.csv' | head -1)" $out/${tag}_bench_data_${data}_mode3_kernel_stats.csv
  python3 $root/tools/vsm_kernel_times.py $out/${tag}_bench_data_${data}_mode3_kernel_stats.csv | grep -E "sift_(query|rescan|verify_kernel|resolve|select)" 
done
