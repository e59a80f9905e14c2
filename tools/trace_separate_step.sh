#!/bin/bash
# Kernel trace of ONE single-utterance pipeline.separate call (EM -> DHTV -> alignment -> PSD -> gev+ban -> apply;
# BASELINE configs[1] shape): every launch of the last call with its duration and start offset.
#   gpurun -- 'bash tools/trace_separate_step.sh [gev+ban|mvdr_souden]'  -> gpurun_out/separate_step.txt
BF=${1:-gev+ban}
R=$(pwd)
cat > /tmp/sep_once.py <<PY
import sys
sys.path.insert(0, '$R')
import torch
from pb_bss_amd import _lib, pipeline
from pb_bss_amd.testing import synth
Y, init = synth.make_stft(513, 500, 8, 3, seed=0)
Yd, gd = _lib.to_device(Y[None]), _lib.to_device(init[None])
for _ in range(6):
    out = pipeline.separate(Yd, gd, iterations=100, stft_size=1024, beamformer='$BF')
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_separate
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_separate -o p -- python /tmp/sep_once.py > $R/gpurun_out/separate_trace.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/separate_step.txt
import csv, glob
f = glob.glob('gpurun_out/prof_separate/**/p_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'cacgmm_em_kernel' in r['Kernel_Name']]
a = idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
print('# the last pipeline.separate call under rocprofv3 --kernel-trace: us | start offset us | kernel')
for r in rows[a:]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print('%8.1f  +%8.1f  %s' % (d / 1e3, (int(r['Start_Timestamp']) - t0) / 1e3, r['Kernel_Name'][:110]))
print('launches', len(rows) - a)
PY
