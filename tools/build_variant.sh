#!/bin/bash
# Build a variant of libofl_hip.so with extra -D switches (A/B of tuning switches on one GPU box via OFL_HIP_LIB):
#   tools/build_variant.sh <name> [-DOFL_...=x ...]   ->  tools/microbench/var/<name>.so   (git-ignored; travels with gpurun)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/tools/microbench/var"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fvisibility=hidden "$@" \
  -I "$root/include" -o "$root/tools/microbench/var/$name.so" "$root/oflibpytorch_amd/csrc/ofl_kernels.hip" "$root/oflibpytorch_amd/csrc/ofl_aux_kernels.hip" "$root/oflibpytorch_amd/csrc/ofl_warp_wide.hip" "$root/oflibpytorch_amd/csrc/ofl_splat_gather.hip"
echo "$root/tools/microbench/var/$name.so"
