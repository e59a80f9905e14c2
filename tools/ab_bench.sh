#!/bin/bash
# A/B two builds of the library on ONE box (box-to-box variance is ~10%):
#   gpurun -- 'bash tools/ab_bench.sh ab_base.so pb_bss_amd/libpbbss_hip.so'
A=${1:?baseline .so}; B=${2:?candidate .so}
for i in 1 2 3; do
  for lib in "$A" "$B"; do
    PBBSS_LIB=$(readlink -f "$lib") python bench.py --steps 40 --warmup 5 --cpu-iters 0 --check-bins 8 --config3 off --configs45 off --extras off --f32 off --sustained-s 0 2>/dev/null | tail -1 | \
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$lib', 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'err %.2e' % d['mask_max_abs_err'], 'status', d['status_bits_or'])"
  done
done
