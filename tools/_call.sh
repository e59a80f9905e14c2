cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_embed.py tests/test_gpu_embed_stepwise.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04f_pytest.log
L=gpurun_out/r04f_variants.log
: > $L
run() { echo "== $*" >> $L; env "$@" timeout 200 python tools/bench_embed.py --joint-only 2>&1 | grep "device" | awk 'NR%3==0' >> $L; }
run PBBSS_X=1
run PBBSS_JOINT_SWEEP_ROWS=128 PBBSS_JOINT_SWEEP_CHUNKS=1024
run PBBSS_JOINT_SWEEP_ROWS=128 PBBSS_JOINT_SWEEP_CHUNKS=2004
run PBBSS_JOINT_SWEEP_ROWS=128 PBBSS_JOINT_SWEEP_CHUNKS=512
run PBBSS_JOINT_SWEEP_ROWS=64 PBBSS_JOINT_SWEEP_CHUNKS=2048
run PBBSS_JOINT_SWEEP_ROWS=64 PBBSS_JOINT_SWEEP_CHUNKS=4008
run BENCH_F=512
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04f_trace -o p -- python $GRAFT_REPO_ROOT/tools/bench_embed.py --joint-only > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' >> $L
import csv,glob
for f in glob.glob('gpurun_out/r04f_trace/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage'])>0.3: print(r['Name'][:100],'|',r['Calls'],'|',round(float(r['AverageNs'])/1e3,2),'|',r['Percentage'])
PY
cat gpurun_out/r04f_pytest.log; cat $L
