#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: an un-profiled run of the headline command, one
# kernel-trace/stats pass of the SAME command, and separate PMC passes (never combined with other
# trace domains).  The command pre-heats the chip (bench.py --preheat-s) and then runs 5 warm-up
# + 25 measured steps; the summary reports median / min of the 25 MEASURED launches only, next to
# the un-profiled HIP-event time of the same box and the profiler's overhead as their ratio.
#   gpurun -- 'bash tools/profile_round.sh r03_a'   -> gpurun_out/r03_a_profile.txt
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
STEPS=25
WARM=5
CMD="python $ROOT/bench.py --steps $STEPS --warmup $WARM --cpu-iters 0 --check-bins 0 --config3 off --configs45 off --f32 off --extras off --sustained-s 0"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_$TAG"
mkdir -p "$OUT/prof_$TAG"
$CMD > "$OUT/prof_$TAG/unprofiled.json" 2> "$OUT/prof_$TAG/unprofiled.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/trace" -o p -- $CMD \
  > "$OUT/prof_$TAG.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/prof_$TAG/pmc_$name" -o p -- \
    $CMD --preheat-s 0.2 >> "$OUT/prof_$TAG.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_csv_summary.py "$OUT/prof_$TAG" "$CMD" "$(python bench.py --print-source-sha)" $STEPS \
  > "$OUT/${TAG}_profile.txt"
cat "$OUT/${TAG}_profile.txt"
