cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04o_pytest.log
timeout 600 bash tools/profile_round.sh r04_e > gpurun_out/r04o_prof2.log 2>&1
timeout 400 bash tools/profile_workload.sh r04_e config5 > gpurun_out/r04o_prof5.log 2>&1
timeout 400 bash tools/profile_workload.sh r04_e config4 watson > gpurun_out/r04o_prof4.log 2>&1
timeout 400 bash tools/profile_workload.sh r04_e config4 vmf > gpurun_out/r04o_prof4v.log 2>&1
cp gpurun_out/r04_e_*profile.txt profiles/ 2>/dev/null
timeout 600 python bench.py > gpurun_out/r04_e_bench.json 2> gpurun_out/r04o_bench.err
cat gpurun_out/r04o_pytest.log; tail -3 gpurun_out/r04o_bench.err; ls gpurun_out/r04_e_*
