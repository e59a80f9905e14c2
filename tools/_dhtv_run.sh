mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_permutation_alignment.py -x -q 2>&1 | tail -15
for team in 0 -16 16 8 4 1; do echo "== team $team"; PBBSS_DHTV_TEAM=$team timeout 300 python tools/bench_alignment.py 2>&1 | grep "device DHTV\|NumPy oracle"; done
