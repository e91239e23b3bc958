#!/bin/bash
# Builds the command-line driver and tools/abi_exerciser.cpp with AddressSanitizer on the HOST side of every source (device code is
# not instrumented) into build/asan/vsm_match_pairs_asan.  Run it on a GPU box with
#   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 build/asan/vsm_match_pairs_asan --sift-dir ... 
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/asan
mkdir -p $out
cd $root/visual_slam_amd/csrc
for f in sift_kernels f32_kernels bf16_kernels f32_match geom_kernels shortlist_kernels vsm_api; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Xarch_host -fsanitize=address \
      -Xarch_host -fno-omit-frame-pointer -c -o $out/$f.o $f.hip
done
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=address -c -o $out/formats.o formats.cpp
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -Xarch_host -fsanitize=address -I$root/include -c -o $out/cli.o match_pairs_cli.cpp
cd $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -pthread -o vsm_match_pairs_asan \
    cli.o sift_kernels.o f32_kernels.o bf16_kernels.o f32_match.o geom_kernels.o shortlist_kernels.o vsm_api.o formats.o
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -Xarch_host -fsanitize=address -I$root/include -c -o $out/abi.o $root/tools/abi_exerciser.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -pthread -o abi_exerciser_asan \
    abi.o sift_kernels.o f32_kernels.o bf16_kernels.o f32_match.o geom_kernels.o shortlist_kernels.o vsm_api.o formats.o
echo "built $out/vsm_match_pairs_asan and $out/abi_exerciser_asan"
