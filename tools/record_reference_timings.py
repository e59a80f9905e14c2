#!/usr/bin/env python3
"""Time the UNMODIFIED reference (fgnt/pb_bss, imported from /root/reference through
oracle/refshim.py) on the BASELINE configurations, on this host's cores, and write
profiles/reference_cpu_timings.json.

/root/reference exists in the build container only -- not on the GPU box where bench.py runs --
so bench.py attaches this committed file to its `cpu_baseline` blocks as `reference_recorded`
(it/s of the reference itself, host named), next to the live timing of the NumPy oracle on the
GPU box's own host.  Every figure is the median of REPEATS runs; all runs are kept.

    python tools/record_reference_timings.py            # ~3-4 minutes of CPU
"""
import json
import os
import platform
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

REPEATS = 3


def timed(fn, iterations):
    runs = []
    for _ in range(REPEATS):
        t0 = time.perf_counter()
        fn()
        runs.append(iterations / (time.perf_counter() - t0))
    return {'it_per_s_median': statistics.median(runs), 'it_per_s_runs': runs,
            'iterations_per_run': iterations, 'repeats': REPEATS}


def vmf_features(Y):
    """Real-valued features of a multi-channel observation for the vMF mixture of config 4:
    phase-normalised to channel 0, (Re, Im) stacked, unit norm -> (F, T, 2D) float64."""
    ph = np.exp(-1j * np.angle(Y[..., :1]))
    z = Y * ph
    f = np.concatenate([z.real, z.imag], axis=-1).astype(np.float64)
    return f / np.maximum(np.linalg.norm(f, axis=-1, keepdims=True), 1e-300)


def main():
    import warnings
    warnings.filterwarnings('ignore')
    from oracle import refshim, cacgmm as oc, cwmm as ow, embed as oe, beamformer as ob
    from pb_bss_amd.testing import synth
    refshim.load()
    import pb_bss.distribution as dist
    import pb_bss.extraction.beamformer as bf

    def shaped(fn):
        with oc.reference_shaped():
            return fn()

    cpu = ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
    except Exception:  # noqa: BLE001
        pass
    out = {
        'what': 'EM iterations/s of the unmodified reference (pb_bss imported from /root/reference) '
                'and of the NumPy oracle on the same host, median of %d runs each' % REPEATS,
        'script': 'tools/record_reference_timings.py',
        'host': {'cpu': cpu, 'logical_cores': os.cpu_count(), 'platform': platform.platform(),
                 'numpy': np.__version__, 'threads_note':
                     'np.einsum (c_einsum) is single-threaded; LAPACK eigh / solve calls on D x D '
                     'matrices do not thread either: 1 core is what the reference uses'},
        'configs': {},
    }

    # ---- BASELINE configs[1]: F=513 T=500 D=8 K=3 cACGMM -----------------------------------
    F, T, D, K = 513, 500, 8, 3
    Y, init = synth.make_stft(F, T, D, K, seed=0)
    Y128 = Y.astype(np.complex128)
    n = 10
    out['configs']['config2'] = {
        'shape': dict(F=F, T=T, D=D, K=K),
        'reference_f64': timed(lambda: dist.CACGMMTrainer().fit(Y128, initialization=init,
                                                               iterations=n), n),
        'reference_f32': timed(lambda: dist.CACGMMTrainer().fit(Y, initialization=init,
                                                               iterations=n), n),
        'oracle_f64': timed(lambda: oc.em_fit(Y128, init, iterations=n), n),
        'oracle_reference_shaped': timed(lambda: shaped(lambda: oc.em_fit(Y128, init, iterations=n)), n),
        'note': 'oracle_reference_shaped: the oracle in its timing mode (oracle/cacgmm.py '
                'REFERENCE_SHAPED: the reference\'s own five-operand einsum(optimize=\'optimal\') in '
                'the E-step) -- what bench.py times as cpu_baseline on the GPU box; reference_f64: complex128 observation (the arithmetic the device kernel is held '
                'to); reference_f32: complex64 observation + ndarray initialisation, the '
                'reference\'s own single-precision path (cacgmm.py:226-227)',
    }
    print(json.dumps(out['configs']['config2']), flush=True)

    # ---- BASELINE configs[3]: F=257 T=800 D=6 K=3, Watson / vMF + MVDR-Souden -----------------
    F, T, D, K = 257, 800, 6, 3
    Y, init = synth.make_stft(F, T, D, K, seed=0)
    Y128 = Y.astype(np.complex128)
    feat = vmf_features(Y128)
    n = 10

    def ref_chain():
        m = dist.CWMMTrainer().fit(Y128, initialization=init, iterations=n)
        masks = m.predict(Y128)
        X = Y128.transpose(0, 2, 1)
        psd = bf.get_power_spectral_density_matrix(X, masks)
        for k in range(K):
            w = bf.get_mvdr_vector_souden(psd[:, k], psd.sum(1) - psd[:, k])
            bf.apply_beamforming_vector(w, X)

    def oracle_chain():
        m = ow.cwmm_fit(Y128, init, iterations=n)
        masks = ow.cwmm_predict(m, Y128)
        X = Y128.transpose(0, 2, 1)
        psd = ob.psd(X, masks)
        for k in range(K):
            ob.apply_bf(ob.mvdr_souden(psd[:, k], psd.sum(1) - psd[:, k]), X)

    out['configs']['config4'] = {
        'shape': dict(F=F, T=T, D=D, K=K),
        'reference_watson_chain': timed(ref_chain, n),
        'oracle_watson_chain': timed(oracle_chain, n),
        'reference_watson_fit': timed(lambda: dist.CWMMTrainer().fit(Y128, initialization=init,
                                                                    iterations=n), n),
        'oracle_watson_fit': timed(lambda: ow.cwmm_fit(Y128, init, iterations=n), n),
        'oracle_watson_chain_reference_shaped': timed(lambda: shaped(oracle_chain), n),
        'oracle_vmf_fit_reference_shaped': timed(lambda: shaped(lambda: oe.vmfmm_fit(feat, init, n)), n),
        'reference_vmf_fit': timed(lambda: dist.VMFMMTrainer().fit(feat, initialization=init,
                                                                  iterations=n), n),
        'oracle_vmf_fit': timed(lambda: oe.vmfmm_fit(feat, init, n), n),
        'note': '*_chain: CWMMTrainer.fit (10 iterations) + predict + PSD + get_mvdr_vector_souden '
                'per class + apply, iterations/s over the whole chain; vmf: VMFMMTrainer on the '
                '(F, T, 2D) real features of tools/record_reference_timings.py:vmf_features, one '
                'mixture per bin',
    }
    print(json.dumps(out['configs']['config4']), flush=True)

    # ---- BASELINE configs[4]: joint GCACGMM, F=513 T=500 D=8 K=3 E=40 -------------------------
    F, T, D, K, E = 513, 500, 8, 3, 40
    Y, e, init = synth.make_joint(F, T, D, K, E, seed=0)
    Y128, e64 = Y.astype(np.complex128), e.astype(np.float64)
    n = 3
    out['configs']['config5'] = {
        'shape': dict(F=F, T=T, D=D, K=K, E=E),
        'reference_gcacgmm_fit': timed(lambda: dist.GCACGMMTrainer().fit(
            Y128, e64, initialization=init, iterations=n), n),
        'oracle_gcacgmm_fit': timed(lambda: oe.joint_fit('gaussian', Y128, e64, init, n), n),
        'oracle_gcacgmm_fit_reference_shaped': timed(
            lambda: shaped(lambda: oe.joint_fit('gaussian', Y128, e64, init, n)), n),
        'note': 'GCACGMMTrainer.fit, spherical Gaussian on the embedding (its default), float64',
    }
    print(json.dumps(out['configs']['config5']), flush=True)

    path = os.path.join(ROOT, 'profiles', 'reference_cpu_timings.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
