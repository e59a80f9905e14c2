#!/bin/bash
# rocprofv3 evidence for BASELINE configs[2] / [3] / [4] (bench.py --workload config3|config4|config5):
# an un-profiled run, one --kernel-trace --stats pass and SEPARATE --pmc passes of the same
# command (never combined with other trace domains).  The step of the profiled command is the EM
# region only (config4: --c4-extraction off), so that the kernels of one step in the trace are the
# ones bench.py's HIP events bracket.
#   gpurun -- 'bash tools/profile_workload.sh r04_a config5'
#   gpurun -- 'bash tools/profile_workload.sh r04_a config4 watson'
#   gpurun -- 'bash tools/profile_workload.sh r05_a config3'        (the whole 64-utterance chain)
#   -> gpurun_out/<tag>_<workload>[_vmf]_profile.txt   (copy to profiles/)
set -u
TAG=${1:-rXX}
WL=${2:-config5}
LEG=${3:-watson}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
STEPS=20
WARM=3
[ "$WL" = config3 ] && STEPS=6 && WARM=2
NAME=$WL
[ "$WL" = config4 ] && [ "$LEG" = vmf ] && NAME=config4_vmf
CMD="python $ROOT/bench.py --workload $WL --leg $LEG --steps $STEPS --warmup $WARM --cpu-iters 0 --check-bins 0 --c4-extraction off --preheat-s 0.3"
D="$OUT/prof_${TAG}_$NAME"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$D"
mkdir -p "$D"
$CMD > "$D/unprofiled.json" 2> "$D/unprofiled.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o p -- $CMD > "$D.log" 2>&1
# PBBSS_PMC_GROUPS="A B;C D": other counter groups (one pass each) instead of the traffic set, e.g. the
# instruction mix  PBBSS_PMC_GROUPS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM;SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
if [ -n "${PBBSS_PMC_GROUPS:-}" ]; then IFS=';' read -ra GROUPS_ <<< "$PBBSS_PMC_GROUPS"; else
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum"); fi
for grp in "${GROUPS_[@]}"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$D/pmc_$name" -o p -- \
    $CMD >> "$D.log" 2>&1
done
cd "$ROOT"
python tools/rocprof_workload_summary.py "$D" "$CMD" \
  "$(python bench.py --workload $WL --leg $LEG --print-source-sha)" $NAME > "$OUT/${TAG}_${NAME}_profile.txt"
cat "$OUT/${TAG}_${NAME}_profile.txt"
