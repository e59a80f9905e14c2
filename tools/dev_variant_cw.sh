#!/bin/bash
# Kernel-development helper for the complex-Watson kernels: build a variant of ONE sensor count's
# translation unit (cw_inst.hip, default D = 6: BASELINE configs[3]) and link it with the other
# objects of the shipped library.
#   tools/dev_variant_cw.sh <tag> [D] [extra hipcc flags...]   -> pb_bss_amd/libpbbss_hip_<tag>.so
# ISA of the variant: /tmp/devcw_<tag>/cw_inst-hip-amdgcn-amd-amdhsa-gfx950.s
set -e
TAG=${1:?tag}; shift
D=6
if [[ "${1:-}" =~ ^[2-8]$ ]]; then D=$1; shift; fi
cd "$(dirname "$0")/../pb_bss_amd/csrc"
mkdir -p /tmp/devcw_$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I. -I../../include"
hipcc $FLAGS -DPBBSS_EM_D=$D "$@" -Rpass-analysis=kernel-resource-usage \
  -save-temps=obj -c cw_inst.hip -o /tmp/devcw_$TAG/cw_d$D.o 2>&1 | grep -E "Function Name|VGPRs:|Scratch|Occupancy" | \
  sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - - - | sed 's/ \+/ /g' | grep "Li3E" || true
OBJS=$(ls build/*.o | grep -v "cw_d$D\.o\|_prof\.o" | tr '\n' ' ')
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpbbss_hip_$TAG.so $OBJS /tmp/devcw_$TAG/cw_d$D.o -ldl
ls -la ../libpbbss_hip_$TAG.so
