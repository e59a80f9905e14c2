"""GPU: error behaviour of the C ABI itself (include/pbbss.h): integer return codes, no
exceptions or aborts across the boundary, handles can be created / destroyed repeatedly,
and calls on separate handles from separate host threads do not interfere."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _env():
    import torch
    from pb_bss_amd import _lib
    lib = _lib.load()
    return torch, _lib, lib, _lib.handle(0), _lib.stream_ptr(0)


def test_invalid_arguments_return_codes_not_crashes():
    torch, _lib, lib, h, stream = _env()
    B, T, D, K = 3, 40, 4, 2
    y = torch.zeros((B, T, D), dtype=torch.complex64, device='cuda')
    g = torch.full((B, K, T), 0.5, dtype=torch.float64, device='cuda')
    vec = torch.zeros((B, K, D, D), dtype=torch.complex128, device='cuda')
    val = torch.zeros((B, K, D), dtype=torch.float64, device='cuda')
    w = torch.zeros((B, K), dtype=torch.float64, device='cuda')
    st = torch.zeros((B, K), dtype=torch.int32, device='cuda')
    opts = _lib.EmOpts(iterations=2, covariance_norm=1, weight_mode=0, layout=0,
                       affiliation_eps=1e-10, eigenvalue_floor=1e-10)

    def fit(handle=h, yy=y, gg=g, KK=K, DD=D, o=opts, vv=vec):
        return lib.pbbss_cacgmm_fit(handle, _lib.ptr(yy), B, T, DD, KK, _lib.ptr(gg), None, None,
                                    None, None, None, ctypes.byref(o) if o else None, _lib.ptr(vv),
                                    _lib.ptr(val), _lib.ptr(w), _lib.ptr(st), None, None, stream)
    assert fit(handle=None) == _lib.ERR_INVALID_ARG
    assert fit(yy=None) == _lib.ERR_INVALID_ARG
    assert fit(o=None) == _lib.ERR_INVALID_ARG
    assert fit(vv=None) == _lib.ERR_INVALID_ARG
    assert fit(gg=None) == _lib.ERR_INVALID_ARG              # neither affiliations nor a model
    assert fit(DD=35) == _lib.ERR_UNSUPPORTED              # 2..8 fused kernel, 9..34 generic path
    assert fit(KK=20) == _lib.ERR_UNSUPPORTED             # 1..6 fused kernel, 7..19 generic path
    bad = _lib.EmOpts(iterations=0, covariance_norm=1)
    assert fit(o=bad) == _lib.ERR_INVALID_ARG                # cacgmm.py:200 asserts iterations > 0
    bad = _lib.EmOpts(iterations=1, covariance_norm=7)
    assert fit(o=bad) == _lib.ERR_INVALID_ARG
    for code in (0, -1, -2, -3, -4, -99):
        assert isinstance(lib.pbbss_error_string(code), bytes)
    # embedding entry points: E > 256 and K > 6 are refused, not mis-executed
    e = torch.zeros((1, 10, 300), dtype=torch.float32, device='cuda')
    m = torch.zeros((1, 2, 300), dtype=torch.float64, device='cuda')
    s = torch.ones((1, 2), dtype=torch.float64, device='cuda')
    o = torch.zeros((1, 2, 10), dtype=torch.float64, device='cuda')
    assert lib.pbbss_embed_log_pdf(h, _lib.ptr(e), 0, 1, 10, 300, 2, 0, _lib.ptr(m), _lib.ptr(s),
                                   _lib.ptr(o), stream) == _lib.ERR_UNSUPPORTED
    # LCMV needs K <= D
    a = torch.zeros((3, 5, 2), dtype=torch.complex128, device='cuda')
    assert lib.pbbss_lcmv(h, _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), 5, 2, 3, _lib.ptr(a), None,
                          stream) == _lib.ERR_INVALID_ARG
    # condition_covariance works entry by entry: in place is refused, not raced
    cx = torch.zeros((3, 4, 4), dtype=torch.complex128, device='cuda')
    co = torch.zeros_like(cx)
    assert lib.pbbss_condition_covariance(h, _lib.ptr(cx), 3, 4, ctypes.c_double(0.1), _lib.ptr(cx),
                                          stream) == _lib.ERR_INVALID_ARG
    assert lib.pbbss_condition_covariance(h, _lib.ptr(cx), 3, 4, ctypes.c_double(0.1), _lib.ptr(co),
                                          stream) == _lib.OK
    # permutation solvers: K <= 8, metric in range, mapping window inside the output row
    mk = torch.rand((1, 9, 5, 16), dtype=torch.float64, device='cuda')
    mp = torch.zeros((1, 9, 5), dtype=torch.int32, device='cuda')
    ps = torch.zeros((1,), dtype=torch.int32, device='cuda')
    i64x3 = ctypes.c_int64 * 3

    def pair(KK=3, metric=0, col0=0, F=5, map_F=5, strides=i64x3(9 * 5 * 16, 5 * 16, 16)):
        return lib.pbbss_pa_pairwise_mapping(h, _lib.ptr(mk), _lib.ptr(mk), 1, KK, F, 16, strides,
                                             strides, metric, 0, None, _lib.ptr(mp), map_F, col0,
                                             _lib.ptr(ps), stream)
    assert pair(KK=9) == _lib.ERR_UNSUPPORTED
    assert pair(metric=3) == _lib.ERR_INVALID_ARG
    assert pair(col0=1) == _lib.ERR_INVALID_ARG             # 1 + 5 columns do not fit 5
    assert pair(strides=None) == _lib.ERR_INVALID_ARG
    assert pair() == _lib.OK
    assert lib.pbbss_pa_compose_mapping(h, None, 1, 3, 5, stream) == _lib.ERR_INVALID_ARG
    assert lib.pbbss_pa_compose_mapping(h, _lib.ptr(mp), 1, 9, 5, stream) == _lib.ERR_UNSUPPORTED
    assert lib.pbbss_pa_mapping_from_scores(h, _lib.ptr(mk), 0, 3, 0, _lib.ptr(mp), _lib.ptr(ps),
                                            stream) == _lib.ERR_INVALID_ARG
    # Gaussian mixture: same argument rules as the vMF mixture (affiliations xor model)
    mo = _lib.MixOpts(iterations=2, kind=_lib.EMBED_GAUSS_SPHERICAL)
    ge = torch.rand((1, 10, 4), dtype=torch.float32, device='cuda')
    gm = torch.zeros((1, 2, 4), dtype=torch.float64, device='cuda')
    gc = torch.ones((1, 2), dtype=torch.float64, device='cuda')
    assert lib.pbbss_gmm_fit(h, _lib.ptr(ge), 1, 10, 4, 2, None, None, None, None, None, None,
                             ctypes.byref(mo), _lib.ptr(gm), _lib.ptr(gc), _lib.ptr(gc), None,
                             None, stream) == _lib.ERR_INVALID_ARG
    torch.cuda.synchronize()
    assert fit() == _lib.OK                                   # the handle is still usable


def test_joint_fit_of_a_long_utterance_is_served_not_refused():
    """Round 3 reported PBBSS_ERR_LDS_CAPACITY for an utterance too long for the LDS-resident
    joint kernels; since round 4 the size-generic kernels take it (the reference has no length
    limit) -- the result is checked against the oracle in tests/test_gpu_embed.py."""
    from oracle import synth
    from pb_bss_amd.distribution import GCACGMMTrainer
    Y, e, init = synth.make_joint(2, 4000, 8, 3, 4, seed=0)
    model = GCACGMMTrainer().fit(Y, e, initialization=init, iterations=2)
    assert model.cacg.covariance_eigenvalues.shape == (2, 3, 8)
    assert np.isfinite(model.cacg.covariance_eigenvalues).all()


def test_handles_create_destroy_and_threads():
    torch, _lib, lib, _, _ = _env()
    for _ in range(20):
        hh = ctypes.c_void_p()
        assert lib.pbbss_create(ctypes.byref(hh), 0) == _lib.OK
        assert lib.pbbss_destroy(hh) == _lib.OK
    assert lib.pbbss_destroy(None) == _lib.ERR_INVALID_ARG
    from oracle import synth
    from pb_bss_amd.distribution import CACGMMTrainer
    Y, init = synth.make_stft(40, 120, 4, 2, seed=1)
    want = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=10)
    got, errs = {}, []

    def work(i):
        try:
            got[i] = CACGMMTrainer().fit_predict(Y, initialization=init, iterations=10)
        except Exception as exc:  # noqa: BLE001 - reported below
            errs.append(exc)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(got[i], want)


def test_souden_binding_of_integration_section_j_through_the_raw_abi():
    """The reference-side binding INTEGRATION.md section J shows for pbbss_mvdr_souden ->
    pbbss_select_reference_channel -> pbbss_apply_beamforming_vector_shared, written out with
    plain ctypes on device pointers: same beam-forming vectors and outputs as the oracle's
    get_mvdr_vector_souden (automatic reference channel) + apply per class; bad arguments of the
    two round-6 exports come back as codes."""
    torch, _lib, lib, h, stream = _env()
    from oracle import beamformer as ob
    rng = np.random.default_rng(21)
    K, F, D, T = 3, 33, 5, 70

    def psd():
        a = rng.standard_normal((K, F, D, 2 * D)) + 1j * rng.standard_normal((K, F, D, 2 * D))
        return a @ a.conj().swapaxes(-1, -2) / (2 * D)
    target, noise = psd(), psd() + 0.1 * np.eye(D)
    X = (rng.standard_normal((F, D, T)) + 1j * rng.standard_normal((F, D, T))).astype(np.complex64)
    eps = float(np.finfo(np.float64).tiny)
    td, nd, xd = (_lib.to_device(a) for a in (target, noise, X))
    c128 = dict(dtype=torch.complex128, device='cuda')
    mat, num, den = (torch.empty(s, **c128) for s in ((K * F, D, D), (K * F, D), (K * F, D)))
    st = torch.empty(K * F, dtype=torch.int32, device='cuda')
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    assert lib.pbbss_mvdr_souden(h, p(td), p(nd), K * F, D, ctypes.c_double(eps), p(mat), p(num),
                                 p(den), p(st), stream) == 0
    w = torch.empty((K, F, D), **c128)
    ref, ok = (torch.empty(K, dtype=torch.int32, device='cuda') for _ in range(2))
    sel = lambda **kw: lib.pbbss_select_reference_channel(  # noqa: E731
        kw.get('h', h), kw.get('mat', p(mat)), p(num), p(den), kw.get('L', K), F, kw.get('D', D), F, 1,
        ctypes.c_double(eps), kw.get('w', p(w)), p(ref), p(ok), stream)
    assert sel() == 0
    s = torch.empty((K, F, T), **c128)
    app = lambda B=K * F, xb=F: lib.pbbss_apply_beamforming_vector_shared(  # noqa: E731
        h, p(w), p(xd), 0, B, xb, T, D, p(s), stream)
    assert app() == 0
    assert bool(ok.all())
    got_w, got_s = _lib.to_host(w), _lib.to_host(s)
    for k in range(K):
        want = ob.mvdr_souden(target[k], noise[k])
        np.testing.assert_allclose(got_w[k], want, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(got_s[k], ob.apply_bf(want, X.astype(np.complex128)),
                                   rtol=1e-9, atol=1e-10)
    assert sel(h=None) == _lib.ERR_INVALID_ARG
    assert sel(L=0) == _lib.ERR_INVALID_ARG
    assert sel(w=p(mat)) == _lib.ERR_INVALID_ARG            # out_w must not alias the matrices
    assert sel(D=33) == _lib.ERR_UNSUPPORTED
    assert app(B=K * F + 1) == _lib.ERR_INVALID_ARG         # B is not a multiple of x_batch
    assert app(xb=0) == _lib.ERR_INVALID_ARG
