cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_embed.py tests/test_gpu_embed_stepwise.py tests/test_gpu_properties.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04p_pytest.log
timeout 300 python tools/bench_embed.py --no-cpu 2>&1 | grep device > gpurun_out/r04p_embed.log
cat gpurun_out/r04p_pytest.log gpurun_out/r04p_embed.log
