#!/bin/bash
# Kernel trace of the full-covariance Gaussian path (tools/bench_gauss_full.py);
# summary -> gpurun_out/gauss_full_kernels.txt (copy into profiles/ to keep it).
out=$GRAFT_REPO_ROOT/gpurun_out/gauss_full_kernels.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pgf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgf -o p -- \
  python $GRAFT_REPO_ROOT/tools/bench_gauss_full.py --no-cpu > /tmp/pgf.log 2>&1
f=$(find /tmp/pgf -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats of: python tools/bench_gauss_full.py --no-cpu  (N=256500 E=40 K=3)"
  grep "us\b\|us/iter\|TFLOP" /tmp/pgf.log
  echo "# name | calls | avg_us | min_us | max_us | pct"
  python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:120], "|", r["Calls"], "| %.1f | %.1f | %.1f |"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3), r["Percentage"])
PY
} > $out
cat $out
