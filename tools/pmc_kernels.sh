#!/bin/bash
# Issue-slot evidence for an arbitrary timing script: two separate rocprofv3 --pmc passes
# (never combined with other trace domains) + a kernel trace, summarised per kernel of the
# library.   gpurun -- 'bash tools/pmc_kernels.sh r04_h_generic tools/bench_generic.py --no-cpu'
#   -> gpurun_out/<tag>_pmc.txt
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$TAG/trace -o p -- python $ROOT/"$@" > /tmp/pk_$TAG.log 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pk_$TAG/pmc_$name -o p -- python $ROOT/"$@" >> /tmp/pk_$TAG.log 2>&1
done
cd "$ROOT"
python - "$TAG" "$*" > "$OUT/${TAG}_pmc.txt" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
tag, cmd = sys.argv[1], sys.argv[2]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    m = re.match(r'(pbbss::[A-Za-z0-9_]+(<[^(]*>)?)', n); return m.group(1) if m else n[:60]
print(f'# rocprofv3 kernel trace + two separate --pmc passes of: python {cmd}')
dur, calls = defaultdict(float), defaultdict(int)
for f in glob.glob(f'/tmp/pk_{tag}/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pbbss' in r['Kernel_Name']:
            k = short(r['Kernel_Name']); dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; calls[k] += 1
acc = defaultdict(list)
for f in glob.glob(f'/tmp/pk_{tag}/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pbbss' in r['Kernel_Name']:
            acc[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
tot = sum(dur.values())
print('# SQ_* counters: mean per launch, summed over the waves (quad-cycles); valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES '
      'per wave; issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES')
print('# kernel | launches | avg us | share of library time | waves | VALU instr per wave | valu_busy | issue_stall | parked | lds_busy')
for k in sorted(dur, key=lambda k: -dur[k])[:12]:
    g = lambda c: (sum(acc[(k, c)]) / len(acc[(k, c)])) if acc.get((k, c)) else float('nan')
    wc, waves = g('SQ_WAVE_CYCLES'), g('SQ_WAVES')
    print(f"{k} | {calls[k]} | {dur[k]/calls[k]:.1f} | {100*dur[k]/tot:.1f} % | {waves:.0f} | {g('SQ_INSTS_VALU')/waves:.0f} | "
          f"{g('SQ_ACTIVE_INST_VALU')/wc:.2f} | {g('SQ_WAIT_INST_ANY')/wc:.2f} | {g('SQ_WAIT_ANY')/wc:.2f} | {g('SQ_ACTIVE_INST_LDS')/wc:.2f}")
PY
cat "$OUT/${TAG}_pmc.txt"
