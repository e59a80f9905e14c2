#!/bin/bash
# Kernel trace of the embedding mixtures / joint models (tools/bench_embed.py) on the GPU box;
# the summary lands in gpurun_out/embed_kernels.txt (copy it into profiles/ to keep it).
out=$GRAFT_REPO_ROOT/gpurun_out/embed_kernels.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pe
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o p -- \
  python $GRAFT_REPO_ROOT/tools/bench_embed.py --no-cpu > /tmp/pe.log 2>&1
f=$(find /tmp/pe -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats of: python tools/bench_embed.py --no-cpu"
  echo "# (3 x 100 iterations each of: vMFMM N=256500 E=40 K=3 float32 in; GCACGMM and VMFCACGMM at BASELINE config 5: F=513 T=500 D=8 K=3 E=40)"
  grep "^device" /tmp/pe.log
  echo "# name | calls | avg_us | min_us | max_us | pct"
  python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:110], "|", r["Calls"], "| %.1f | %.1f | %.1f |"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3), r["Percentage"])
PY
} > $out
cat $out
