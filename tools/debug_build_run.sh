#!/bin/bash
# Evidence run of the debug-assert build (make -C pb_bss_amd/csrc -j8 debug) on a GPU box:
# the whole GPU suite with PBBSS_LIB pointing at libpbbss_hip_debug.so (device-side checks of LDS
# carve-ups, frame and slab indices; -O1).  The time-out tests are excluded (they provoke bounded
# waits with a spin limit calibrated for the release build's speed).
#   gpurun -- 'bash tools/debug_build_run.sh r05'   -> gpurun_out/<tag>_debug_build.txt (copy to profiles/)
TAG=${1:-rXX}
OUT=gpurun_out/${TAG}_debug_build.txt
mkdir -p gpurun_out
LIB=pb_bss_amd/libpbbss_hip_debug.so
{
  echo "# command: PBBSS_LIB=libpbbss_hip_debug.so python -m pytest tests -m gpu -q --deselect tests/test_gpu_timeouts.py"
  echo "# library: $LIB  sha256 $(sha256sum $LIB | cut -c1-16)  bytes $(stat -c %s $LIB)"
  echo "# kernel_source_sha (bench.py --print-source-sha): $(python bench.py --print-source-sha)"
  echo "# git HEAD at the time of the run is recorded by the commit that adds this file"
  PBBSS_LIB=libpbbss_hip_debug.so timeout 1500 python -m pytest tests -m gpu -q -rf --deselect tests/test_gpu_timeouts.py 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | head -20
} > $OUT 2>&1
cat $OUT
