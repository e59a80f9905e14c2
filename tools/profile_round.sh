#!/bin/bash
# Collect the rocprofv3 evidence for profiles/: one kernel-trace/stats pass and separate PMC
# passes (never combined with other trace domains) over the headline bench command.
#   gpurun -- 'bash tools/profile_round.sh r01_d'   -> gpurun_out/<tag>_{kernel_trace,pmc_summary}.txt
set -u
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
CMD="python $ROOT/bench.py --steps 6 --warmup 2 --cpu-iters 0 --check-bins 0"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_$TAG"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/trace" -o p -- $CMD \
  > "$OUT/prof_$TAG.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/prof_$TAG/pmc_$name" -o p -- $CMD \
    >> "$OUT/prof_$TAG.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_csv_summary.py "$OUT/prof_$TAG" "$CMD" "$(python bench.py --print-source-sha)" > "$OUT/${TAG}_profile.txt"
cat "$OUT/${TAG}_profile.txt"
