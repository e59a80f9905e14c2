#!/bin/bash
# A/B/C... several builds of the library on ONE box (box-to-box variance is ~10%), interleaved:
#   gpurun -- 'bash tools/ab_multi.sh 3 pb_bss_amd/libpbbss_hip_a.so pb_bss_amd/libpbbss_hip_b.so,PBBSS_OCC2=0 ...'
# (an argument is a library path optionally followed by ,VAR=value environment settings)
ROUNDS=${1:?rounds}; shift
BINS=${BINS:-513}
for i in $(seq $ROUNDS); do
  for spec in "$@"; do
    lib=${spec%%,*}; envs=$(echo "${spec#$lib}" | tr ',' ' ')
    env $envs PBBSS_LIB=$(readlink -f "$lib") python bench.py --steps 40 --warmup 5 --cpu-iters 0 --check-bins 8 --config3 off --configs45 off --extras off --f32 off --sustained-s 0 2>/dev/null | tail -1 | \
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-50s' % '$spec', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'err %.2e' % d['mask_max_abs_err'], 'status', d['status_bits_or'])"
  done
done
