#!/bin/bash
# Matrix-core utilisation of the float kernels: one rocprofv3 --pmc pass each over the database query
# (tools/db_query_probe.py) and the float pair lists (tools/f32_pairs_probe.py), folded by tools/pmc_fold.py.
# usage: tools/bf16_counters.sh TAG     (run on the GPU box; results land in gpurun_out/prof/)
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
counters="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
pass() {   # name, probe script and its arguments
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -s KILL 240 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 "$@" > /tmp/pmc_$name.log 2>&1 || { tail -5 /tmp/pmc_$name.log; return 1; }
  cp "$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)" $out/${tag}_${name}_pmc_mfma.csv
  python3 $root/tools/pmc_fold.py $out/${tag}_${name}_pmc_mfma.csv bf16_ f32_ | tee $out/${tag}_${name}_mfma_utilisation.txt
}
pass dbquery $root/tools/db_query_probe.py && pass f32pairs $root/tools/f32_pairs_probe.py
# the word assignment of the pair shortlist (the tile-minimum kernel with every row's group resolved in its tail)
counters="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
name=shortlist
rm -rf /tmp/pmc_$name
timeout -s KILL 240 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $root/tools/vocab_probe.py --iterations 0 > /tmp/pmc_$name.log 2>&1 || { tail -5 /tmp/pmc_$name.log; exit 1; }
cp "$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)" $out/${tag}_${name}_pmc_mfma.csv
python3 $root/tools/pmc_fold.py $out/${tag}_${name}_pmc_mfma.csv sift_query shortlist | tee $out/${tag}_${name}_mfma_utilisation.txt
# one float pair per call: latency of the call, and what it queues (last three kernels of a kernel trace)
timeout -k 10 120 python3 $root/tools/pair_call_probe.py > $out/${tag}_pair_call_probe.log 2>&1 || exit 1
rm -rf /tmp/tl_pair
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_pair -- python3 $root/tools/pair_call_probe.py > /tmp/tl_pair.log 2>&1 || exit 1
python3 $root/tools/kernel_timeline.py "$(find /tmp/tl_pair -name '*kernel_trace.csv' | head -1)" 6 > $out/${tag}_single_pair_kernel_timeline.txt
cat $out/${tag}_pair_call_probe.log $out/${tag}_single_pair_kernel_timeline.txt
