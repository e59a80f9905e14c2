#!/bin/bash
# Kernel trace of tools/bench_api_sweep.py (PBBSS_SWEEP=2: bench_api_sweep2.py); summary in
# gpurun_out/api_sweep.txt (api_sweep2.txt).
n=${PBBSS_SWEEP:-}
out=$GRAFT_REPO_ROOT/gpurun_out/api_sweep$n.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o p -- \
  python $GRAFT_REPO_ROOT/tools/bench_api_sweep$n.py > /tmp/p4.log 2>&1
f=$(find /tmp/p4 -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats of: python tools/bench_api_sweep$n.py"
  grep -E "ms per call|FAILED|Error|error" /tmp/p4.log
  echo "# kernels by average duration: name | calls | avg_us | min_us | max_us"
  python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["AverageNs"]))
for r in rows[:70]:
    print(r["Name"][:110], "|", r["Calls"], "| %.1f | %.1f | %.1f"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
} > $out
cat $out
