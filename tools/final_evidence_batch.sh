#!/bin/bash
# The evidence of a round in ONE gpurun call (about 12 GPU-minutes), to be run on the FINAL kernel sources -- bench.py only
# attaches PMC traffic from a profile whose header carries the hash of the sources in the tree:
#   gpurun --timeout 2400 -- 'bash tools/final_evidence_batch.sh'
# then copy gpurun_out/r05_p_*_profile.txt, r05_scaling_model.json, r05_debug_build.txt to profiles/ and run bench.py.
# (make -C pb_bss_amd/csrc debug first: the last step runs the GPU suite on libpbbss_hip_debug.so.)
bash tools/profile_round.sh r05_p > gpurun_out/r05p_headline.log 2>&1
bash tools/profile_workload.sh r05_p config3 > gpurun_out/r05p_c3.log 2>&1
bash tools/profile_workload.sh r05_p config4 watson > gpurun_out/r05p_c4.log 2>&1
bash tools/profile_workload.sh r05_p config4 vmf > gpurun_out/r05p_c4v.log 2>&1
bash tools/profile_workload.sh r05_p config5 > gpurun_out/r05p_c5.log 2>&1
python tools/scaling_model.py > gpurun_out/r05_scaling_model.json 2> gpurun_out/r05p_scaling.err
bash tools/debug_build_run.sh r05 > /dev/null 2>&1
tail -3 gpurun_out/r05_debug_build.txt | head -1
ls gpurun_out/*r05_p*
