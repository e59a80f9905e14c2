cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cwmm.py tests/test_gpu_golden.py tests/test_gpu_properties.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04g_pytest.log
L=gpurun_out/r04g_cwmm.log
: > $L
timeout 200 python tools/bench_cwmm.py 2>&1 | grep -v amdgpu >> $L
timeout 300 python bench.py --workload config4 --steps 10 --warmup 2 --cpu-iters 0 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 chain', b['value'], 'em_only', b['em_only'], 'verify', b['verify']['mask_max_abs_err'], b['verify']['ok'])" >> $L
cat gpurun_out/r04g_pytest.log; cat $L
