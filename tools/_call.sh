cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cwmm.py tests/test_gpu_golden.py tests/test_gpu_timeouts.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04q_pytest.log
timeout 300 python tools/_dbg.py 2>&1 | grep -v amdgpu > gpurun_out/r04q_dbg.log
cat gpurun_out/r04q_pytest.log gpurun_out/r04q_dbg.log
