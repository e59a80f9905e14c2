set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a_pytest.log
timeout 600 python bench.py > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
tail -c 600 gpurun_out/r04a_bench.err
PBBSS_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --config3-steps 2 --utterances 4 --preheat-s 0.1 --sustained-s 0 --f32 off > gpurun_out/r04a_bench_n2_rehearsal.json 2> gpurun_out/r04a_bench_n2.err
tail -c 600 gpurun_out/r04a_bench_n2.err
timeout 500 bash tools/profile_workload.sh r04a config5 > gpurun_out/r04a_prof5.log 2>&1
timeout 500 bash tools/profile_workload.sh r04a config4 watson > gpurun_out/r04a_prof4.log 2>&1
cat gpurun_out/r04a_pytest.log
