cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_embed.py tests/test_gpu_embed_stepwise.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r04m_pytest.log
timeout 300 python bench.py --workload config4 --leg vmf --steps 10 --warmup 2 --cpu-iters 0 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 vmf chain', b['value'], 'em_only', b['em_only'], 'verify', b['verify']['mask_max_abs_err'], b['verify']['ok'])" > gpurun_out/r04m_vmf.log
cat gpurun_out/r04m_pytest.log gpurun_out/r04m_vmf.log
