cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_timeouts.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r04h_timeouts.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04h_pytest.log
timeout 600 python bench.py > gpurun_out/r04h_bench.json 2> gpurun_out/r04h_bench.err
L=gpurun_out/r04h_tail.log
: > $L
for w in 64 128 256; do echo "== PBBSS_SPLIT_WINDOW=$w" >> $L; PBBSS_SPLIT_WINDOW=$w timeout 200 python bench.py --steps 30 --warmup 5 --cpu-iters 0 --check-bins 0 --config3 off --configs45 off --f32 off --extras off --sustained-s 0 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'], b['roofline']['kernel_ms'])" >> $L; done
cat gpurun_out/r04h_timeouts.log; cat gpurun_out/r04h_pytest.log; tail -c 300 gpurun_out/r04h_bench.err; cat $L
