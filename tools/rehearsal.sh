#!/bin/bash
# The multi-rank code path of bench.py on a ONE-GPU box: two ranks, both on device 0, collectives over gloo.  The
# values say nothing about scaling; the lines show that sharding, the all-gather + device merge, the max-over-ranks
# timing and the oracle check run.   usage: tools/rehearsal.sh TAG
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
cd $root
export VSM_BENCH_REHEARSAL=1
timeout -k 10 300 python3 bench.py --gpus 2 --images 200 --no-cpu-baseline > $out/${tag}_rehearsal_two_ranks_default.json.log 2>/tmp/reh1.err || { tail -5 /tmp/reh1.err; exit 1; }
timeout -k 10 300 python3 bench.py --gpus 2 --workload db_query --no-cpu-baseline > $out/${tag}_rehearsal_two_ranks_db_query.json.log 2>/tmp/reh2.err || { tail -5 /tmp/reh2.err; exit 1; }
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --images 200 --no-cpu-baseline > $out/${tag}_rehearsal_two_ranks_torchrun.json.log 2>/tmp/reh3.err || { tail -5 /tmp/reh3.err; exit 1; }
for f in default db_query torchrun; do tail -c 420 $out/${tag}_rehearsal_two_ranks_$f.json.log; echo; done
