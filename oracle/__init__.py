"""CPU oracle for the pb_bss cACGMM / beamformer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pb_bss_amd/`` may import this
package; it is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``.

The modules here restate, in plain NumPy, the algorithm of the reference
(fgnt/pb_bss, mounted at /root/reference when available).  Every function
cites the reference ``file:line`` it follows.  The restatement is pinned
against the real reference by ``oracle/make_golden.py`` (run in the build
container, where /root/reference exists) which writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` re-checks the restatement against those
vectors everywhere, including on the GPU box where the reference is absent.
"""
