cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pg && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o p -- python $GRAFT_REPO_ROOT/tools/bench_generic.py --no-cpu > /tmp/pg.log 2>&1; f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1); grep "^D=" /tmp/pg.log; python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:90], "|", r["Calls"], "| %.1f | %.1f | %.1f |"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3), r["Percentage"])
PY
