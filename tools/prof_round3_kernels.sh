#!/bin/bash
# Kernel trace of the kernels that are new in round 3 (tools/bench_round3_kernels.py); the summary
# lands in gpurun_out/round3_kernels.txt (copy it into profiles/ to keep it).
out=$GRAFT_REPO_ROOT/gpurun_out/round3_kernels.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o p -- \
  python $GRAFT_REPO_ROOT/tools/bench_round3_kernels.py > /tmp/p3.log 2>&1
f=$(find /tmp/p3 -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats of: python tools/bench_round3_kernels.py"
  grep "ms per call" /tmp/p3.log
  echo "# name | calls | avg_us | min_us | max_us | pct"
  python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:40]:
    n=r["Name"]
    if n.startswith("void at::") or "rocclr" in n or "rocprim" in n: continue
    print(n[:120], "|", r["Calls"], "| %.1f | %.1f | %.1f |"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3), r["Percentage"])
PY
} > $out
cat $out
