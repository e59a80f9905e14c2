#!/bin/bash
# The evidence of a round in ONE gpurun call (about 12 GPU-minutes), to be run on the FINAL kernel sources -- bench.py only
# attaches PMC traffic from a profile whose header carries the hash of the sources in the tree:
#   gpurun --timeout 2400 -- 'bash tools/final_evidence_batch.sh r06_p'
# then copy gpurun_out/<tag>_*_profile.txt, <round>_scaling_model.json, <round>_debug_build.txt to profiles/ and run bench.py.
# (make -C pb_bss_amd/csrc debug first: the last step runs the GPU suite on libpbbss_hip_debug.so.)
TAG=${1:-r06_p}
RND=${TAG%%_*}
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_headline.log 2>&1
bash tools/profile_workload.sh $TAG config3 > gpurun_out/${TAG}_c3.log 2>&1
bash tools/profile_workload.sh $TAG config4 watson > gpurun_out/${TAG}_c4.log 2>&1
bash tools/profile_workload.sh $TAG config4 vmf > gpurun_out/${TAG}_c4v.log 2>&1
bash tools/profile_workload.sh $TAG config5 > gpurun_out/${TAG}_c5.log 2>&1
python tools/scaling_model.py > gpurun_out/${RND}_scaling_model.json 2> gpurun_out/${TAG}_scaling.err
[ -f pb_bss_amd/libpbbss_hip_debug.so ] && bash tools/debug_build_run.sh $RND > /dev/null 2>&1
tail -3 gpurun_out/${RND}_debug_build.txt 2>/dev/null | head -1
ls gpurun_out/*${TAG}* gpurun_out/${RND}_*
