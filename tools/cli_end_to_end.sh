#!/bin/bash
# End-to-end runs of vsm_match_pairs on 1000 synthetic .sift files of 8192 features (run on the GPU box; logs land
# in gpurun_out/prof/).   usage: tools/cli_end_to_end.sh TAG
set -e
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof
mkdir -p $out
dir=/tmp/sift1000
cli=$root/visual_slam_amd/vsm_match_pairs
python3 $root/tools/make_sift_dir.py $dir --images 1000 --features 8192 > /dev/null
{
  timeout -k 10 200 $cli --sift-dir $dir --num 1000 --out /dev/null 2>&1 | tail -1
  timeout -k 10 200 $cli --sift-dir $dir --num 1000 --out /tmp/matching.txt 2>&1 | tail -1
  ls -l /tmp/matching.txt | awk '{print $5, "bytes matching.txt"}'
  md5sum /tmp/matching.txt | cut -d' ' -f1
} > $out/${tag}_cli_cfg3_end_to_end.log
{
  timeout -k 10 200 $cli --sift-dir $dir --num 1000 --out /tmp/matching.txt --verify 2.0 --inlier-out /tmp/verified.txt 2>&1 | tail -2
  timeout -k 10 200 $cli --sift-dir $dir --num 1000 --out /dev/null --verify 2.0 --inlier-out /dev/null 2>&1 | tail -2
  timeout -k 10 200 $cli --sift-dir $dir --num 1000 --out /dev/null --verify 2.0 --inlier-out /dev/null --camera 1200,1200,800,600 2>&1 | tail -3
} > $out/${tag}_cli_cfg3_verify_end_to_end.log
{
  timeout -k 10 200 $cli --sift-dir $dir --num 1000 --pairs vocab --topk 20 --out /tmp/m2.txt 2>&1 | tail -2
  timeout -k 10 200 $cli --sift-dir $dir --num 1000 --pairs vocab --topk 20 --out /tmp/m2.txt --verify 2.0 --inlier-out /tmp/v2.txt 2>&1 | tail -2
} > $out/${tag}_cli_cfg3_vocab_pairs_end_to_end.log
{
  # two contexts in one process (both on GPU 0 here), one ordered writer: the same file as the one-context run
  md5sum /tmp/matching.txt | cut -d' ' -f1
  timeout -k 10 300 $cli --sift-dir $dir --num 1000 --out /tmp/matching2.txt --devices 0,0 2>&1 | tail -1
  md5sum /tmp/matching2.txt | cut -d' ' -f1
} > $out/${tag}_cli_cfg3_two_contexts_end_to_end.log
{
  # the sliced map-database query through the C++ host (RCCL all-gather with one rank on this box): one slice, eight slices
  timeout -k 10 300 $root/visual_slam_amd/vsm_query_db --synthetic 16384,1000000,256 --slices 1 --reps 5 2>&1 | grep "queries x"
  timeout -k 10 300 $root/visual_slam_amd/vsm_query_db --synthetic 16384,1000000,256 --slices 8 --reps 5 2>&1 | grep "queries x"
} > $out/${tag}_cli_query_db.log
rm -rf $dir /tmp/matching.txt /tmp/matching2.txt /tmp/verified.txt /tmp/m2.txt /tmp/v2.txt
cat $out/${tag}_cli_cfg3_*.log
