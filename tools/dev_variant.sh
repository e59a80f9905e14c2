#!/bin/bash
# Kernel-development helper: build a variant of the D=8 EM translation unit (K=3, complex64 only:
# 10 s instead of 2 min) and link it with the other objects of the shipped library.
#   tools/dev_variant.sh <tag> [extra hipcc flags...]   -> pb_bss_amd/libpbbss_hip_<tag>.so
# ISA of the variant: /tmp/dev_<tag>/em_inst-hip-amdgcn-amd-amdhsa-gfx950.s (tools/isa_blocks.py)
# A/B on one box:  gpurun -- 'bash tools/ab_bench.sh pb_bss_amd/libpbbss_hip_a.so pb_bss_amd/libpbbss_hip_b.so'
set -e
TAG=${1:?tag}; shift
cd "$(dirname "$0")/../pb_bss_amd/csrc"
mkdir -p /tmp/dev_$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I. -I../../include"
hipcc $FLAGS -DPBBSS_EM_D=8 -DPBBSS_EM_DEV_ONLY_K=3 "$@" -Rpass-analysis=kernel-resource-usage \
  -save-temps=obj -c em_inst.hip -o /tmp/dev_$TAG/em_d8.o 2>&1 | grep -E "Function Name|VGPRs:|Scratch|Spill" | \
  sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - - - - | sed 's/ \+/ /g'
OBJS=$(ls build/*.o | grep -v 'em_d8\|_prof\.o' | tr '\n' ' ')
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpbbss_hip_$TAG.so $OBJS /tmp/dev_$TAG/em_d8.o
ls -la ../libpbbss_hip_$TAG.so
