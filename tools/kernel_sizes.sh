#!/bin/bash
# tools/kernel_sizes.sh — every device kernel instance of libhj.so with its code size (bytes of gfx950 ISA), largest first: the list a
# prune starts from.  Runs in the build container (no GPU): each kernel file is compiled device-only, the gfx950
# code object unbundled, its FUNC symbols listed.
cd "$(dirname "$0")/../icde2019-gpu-join_amd/csrc"
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for f in hj_part hj_join hj_util hj_dist; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. --cuda-device-only -c $f.hip -o $T/$f.o 2>/dev/null
  $B/clang-offload-bundler --unbundle --type=o --input=$T/$f.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$f.elf 2>/dev/null || \
  $B/clang-offload-bundler --unbundle --type=o --input=$T/$f.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$T/$f.elf
  $B/llvm-readelf --dyn-syms --wide $T/$f.elf | awk -v F=$f '$4=="FUNC" && $3>0 {print $3, F".hip", $8}'
done | sort -rn | while read sz f sym; do printf "%8d  %-12s %s\n" $sz $f "$(echo $sym | c++filt | cut -c1-150)"; done
rm -rf $T
