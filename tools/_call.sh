cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r04l_gate.log
: > $L
PBBSS_RESIDENCY_GATE=0 timeout 300 python tools/coop_contention_probe.py 16 2>&1 | grep RESIDENCY >> $L
PBBSS_RESIDENCY_GATE=1 timeout 300 python tools/coop_contention_probe.py 16 2>&1 | grep RESIDENCY >> $L
PBBSS_RESIDENCY_GATE=1 timeout 300 python tools/coop_contention_probe.py 32 2>&1 | grep RESIDENCY >> $L
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04l_pytest.log
cat $L; cat gpurun_out/r04l_pytest.log
