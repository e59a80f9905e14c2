bash tools/profile_round.sh r05_p > gpurun_out/r05p_headline.log 2>&1
bash tools/profile_workload.sh r05_p config3 > gpurun_out/r05p_c3.log 2>&1
bash tools/profile_workload.sh r05_p config4 watson > gpurun_out/r05p_c4.log 2>&1
bash tools/profile_workload.sh r05_p config4 vmf > gpurun_out/r05p_c4v.log 2>&1
bash tools/profile_workload.sh r05_p config5 > gpurun_out/r05p_c5.log 2>&1
python tools/scaling_model.py > gpurun_out/r05_scaling_model.json 2> gpurun_out/r05p_scaling.err
bash tools/debug_build_run.sh r05 > /dev/null 2>&1
tail -3 gpurun_out/r05_debug_build.txt | head -1
ls gpurun_out/*r05_p*
