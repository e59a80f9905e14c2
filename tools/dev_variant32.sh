#!/bin/bash
# As tools/dev_variant.sh for the packed-FP32 EM translation unit (em32_inst.hip, D = 8).
#   tools/dev_variant32.sh <tag> [extra hipcc flags...]   -> pb_bss_amd/libpbbss_hip_<tag>.so
set -e
TAG=${1:?tag}; shift
cd "$(dirname "$0")/../pb_bss_amd/csrc"
mkdir -p /tmp/dev32_$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I. -I../../include"
hipcc $FLAGS -DPBBSS_EM_D=8 "$@" -c em32_inst.hip -o /tmp/dev32_$TAG/em32_d8.o
OBJS=$(ls build/*.o | grep -v 'em32_d8\.o\|_prof\.o' | tr '\n' ' ')
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpbbss_hip_$TAG.so $OBJS /tmp/dev32_$TAG/em32_d8.o -ldl
ls -la ../libpbbss_hip_$TAG.so
