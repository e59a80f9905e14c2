#!/bin/bash
# Kernel trace of the end-to-end batch pipeline (examples/separate_batch.py, 8 utterances);
# summary -> gpurun_out/pipeline_kernels.txt (copy into profiles/ to keep it).
out=$GRAFT_REPO_ROOT/gpurun_out/pipeline_kernels.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- \
  python $GRAFT_REPO_ROOT/examples/separate_batch.py --utterances 8 > /tmp/pp.log 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats of: python examples/separate_batch.py --utterances 8"
  grep -v "^$" /tmp/pp.log | tail -6
  echo "# name | calls | avg_us | min_us | max_us | pct"
  python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(r["Name"][:110], "|", r["Calls"], "| %.1f | %.1f | %.1f |"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3), r["Percentage"])
PY
} > $out
cat $out
