#!/bin/bash
# Kernel trace of ONE step of bench.py --workload config4 (mixture fit -> PSD -> MVDR-Souden with the automatic
# reference channel -> apply): every launch between two consecutive EM kernels with its duration and start offset.
#   gpurun -- 'bash tools/trace_chain_step.sh [watson|vmf]'  -> gpurun_out/chain_step_<leg>.txt
LEG=${1:-watson}
R=$(pwd)
MARK=cwmm_em_wide_kernel; [ "$LEG" = vmf ] && MARK=vmf_bin_em2_kernel
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_chain_$LEG
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_chain_$LEG -o p -- python $R/bench.py --workload config4 \
  --leg $LEG --steps 10 --warmup 3 --cpu-iters 0 --check-bins 0 --preheat-s 0.2 --extra-file none > $R/gpurun_out/chain_$LEG.log 2>&1
cd $R
python - "$LEG" "$MARK" <<'PY' | tee gpurun_out/chain_step_$LEG.txt
import csv, glob, sys
leg, mark = sys.argv[1:3]
f = glob.glob(f'gpurun_out/prof_chain_{leg}/**/p_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
wins = [(a, b) for a, b in zip(idx, idx[1:]) if b - a > 5]
a, b = wins[len(wins) // 2]
t0 = int(rows[a]['Start_Timestamp'])
print(f'# one step of the {leg} chain under rocprofv3 --kernel-trace (launch gaps are wider than un-profiled): us | start offset us | kernel')
for r in rows[a:b]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('%8.1f  +%8.1f  %s' % (d / 1e3, (int(r['Start_Timestamp']) - t0) / 1e3, r['Kernel_Name'][:100]))
print('launches', b - a, ' span %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
