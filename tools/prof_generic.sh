#!/bin/bash
# Kernel trace of the generic-size path (tools/bench_generic.py) on the GPU box; the summary
# lands in gpurun_out/generic_kernels.txt (copy it into profiles/ to keep it).
out=$GRAFT_REPO_ROOT/gpurun_out/generic_kernels.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o p -- \
  python $GRAFT_REPO_ROOT/tools/bench_generic.py --no-cpu > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats of: python tools/bench_generic.py --no-cpu"
  echo "# (F=513 T=500 K=3; 2 x 20 EM iterations at each of D = 12, 16, 24, 29, then PSD + gev+ban + apply)"
  grep "^D=" /tmp/pg.log
  echo "# name | calls | avg_us | min_us | max_us | pct"
  python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:100], "|", r["Calls"], "| %.1f | %.1f | %.1f |"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3), r["Percentage"])
PY
} > $out
cat $out
