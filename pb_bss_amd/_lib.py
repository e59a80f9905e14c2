"""ctypes binding of libpbbss_hip.so (the C ABI declared in include/pbbss.h).

The product path has NO CPU fallback: if the shared library is missing or a
call fails this module raises.  PyTorch-ROCm is used only as the owner of
device memory and streams; all arithmetic happens inside the HIP library.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get('PBBSS_LIB', 'libpbbss_hip.so'))

# ---- constants mirrored from include/pbbss.h ---------------------------------
OK = 0
ERR_INVALID_ARG = -1
ERR_UNSUPPORTED = -2
ERR_HIP = -3
ERR_LDS_CAPACITY = -4
ERR_INTERNAL = -5

ST_NONFINITE = 1
ST_EIG_NOCONV = 2
ST_FLOORED = 4
ST_SLOWPATH = 8
ST_NOT_POSDEF = 16
ST_SINGULAR = 32

COVNORM = {False: 0, None: 0, 'eigenvalue': 1, 'trace': 2}
WEIGHT_PER_CLASS_MEAN = 0
WEIGHT_UNIFORM = 1
WEIGHT_SHARED_K = 2   # weight_constant_axis=(-3, -1), pbbss_cacgmm_fit_shared only
WEIGHT_SHARED_KT = 3  # weight_constant_axis=(-3,)
LAYOUT_TD = 0
LAYOUT_DT = 1

EXPORTS = (
    'pbbss_version', 'pbbss_error_string', 'pbbss_create', 'pbbss_destroy',
    'pbbss_normalize_observation', 'pbbss_cacgmm_fit', 'pbbss_cacgmm_fit_shared',
    'pbbss_cacgmm_predict',
    'pbbss_cacg_m_step', 'pbbss_heev_batched', 'pbbss_psd', 'pbbss_gev', 'pbbss_gev_general',
    'pbbss_comm_unique_id', 'pbbss_comm_create', 'pbbss_comm_destroy', 'pbbss_comm_info',
    'pbbss_shard_bounds',
    'pbbss_allgather_masks', 'pbbss_allgather_unpack', 'pbbss_estimate_mixture_weight',
    'pbbss_log_pdf_to_affiliation', 'pbbss_log_pdf_to_affiliation_inline_pa',
    'pbbss_solve', 'pbbss_mvdr_souden', 'pbbss_mvdr', 'pbbss_ban',
    'pbbss_apply_beamforming_vector', 'pbbss_apply_beamforming_vector_shared',
    'pbbss_select_reference_channel', 'pbbss_set_timing',
    'pbbss_last_kernel_ms', 'pbbss_kernel_ms_lagged', 'pbbss_set_phase_profile',
    'pbbss_dhtv_calculate_mapping', 'pbbss_apply_mapping', 'pbbss_cwmm_fit',
    'pbbss_wmwf', 'pbbss_set_split_tail', 'pbbss_split_error', 'pbbss_split_reset', 'pbbss_set_spin_limit',
    'pbbss_embed_log_pdf', 'pbbss_embed_fit', 'pbbss_vmfmm_fit', 'pbbss_joint_fit',
    'pbbss_lcmv', 'pbbss_phase_correction', 'pbbss_snr_postfilter',
    'pbbss_reference_channel_terms', 'pbbss_rank_one_approximation', 'pbbss_matvec',
    'pbbss_distortionless_normalization', 'pbbss_zero_degree_normalization',
    'pbbss_condition_covariance', 'pbbss_apply_online_beamforming_vector',
    'pbbss_set_dhtv_team', 'pbbss_set_dhtv_probe', 'pbbss_stft_num_frames', 'pbbss_stft', 'pbbss_istft',
    'pbbss_pa_pairwise_mapping', 'pbbss_pa_compose_mapping', 'pbbss_pa_mapping_from_scores',
    'pbbss_gmm_fit', 'pbbss_gauss_full_fit', 'pbbss_gauss_full_log_pdf',
    'pbbss_gmm_full_fit',
)

EMBED_VMF = 0
EMBED_GAUSS_SPHERICAL = 1
EMBED_GAUSS_FULL = 2
EMBED_GAUSS_DIAG = 3
# weight_constant_axis of the joint models -> PBBSS_JOINT_WEIGHT_*
JOINT_WEIGHT_FK, JOINT_WEIGHT_UNIFORM, JOINT_WEIGHT_K, JOINT_WEIGHT_KT, JOINT_WEIGHT_CONST = range(5)


class EmOpts(ctypes.Structure):
    """struct pbbss_em_opts"""
    _fields_ = [
        ('iterations', ctypes.c_int32),
        ('covariance_norm', ctypes.c_int32),
        ('weight_mode', ctypes.c_int32),
        ('hermitize', ctypes.c_int32),
        ('layout', ctypes.c_int32),
        ('y_is_c128', ctypes.c_int32),
        ('final_predict', ctypes.c_int32),
        ('force_eig', ctypes.c_int32),
        ('affiliation_eps', ctypes.c_double),
        ('eigenvalue_floor', ctypes.c_double),
        ('precision', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
    ]


PRECISION = {'f64': 0, 'float64': 0, None: 0, 'f32': 1, 'float32': 1}


class CwmmOpts(ctypes.Structure):
    """struct pbbss_cwmm_opts"""
    _fields_ = [
        ('iterations', ctypes.c_int32),
        ('weight_mode', ctypes.c_int32),
        ('y_is_c128', ctypes.c_int32),
        ('final_predict', ctypes.c_int32),
        ('n_coef', ctypes.c_int32),
        ('group', ctypes.c_int32),
        ('ev_min', ctypes.c_double),
        ('ev_max', ctypes.c_double),
        ('max_concentration', ctypes.c_double),
    ]


class MixOpts(ctypes.Structure):
    """struct pbbss_mix_opts"""
    _fields_ = [
        ('iterations', ctypes.c_int32),
        ('kind', ctypes.c_int32),
        ('weight_mode', ctypes.c_int32),
        ('embedding_is_f64', ctypes.c_int32),
        ('obs_is_c128', ctypes.c_int32),
        ('final_predict', ctypes.c_int32),
        ('inline_pa', ctypes.c_int32),
        ('covariance_norm', ctypes.c_int32),
        ('min_concentration', ctypes.c_double),
        ('max_concentration', ctypes.c_double),
        ('affiliation_eps', ctypes.c_double),
        ('eigenvalue_floor', ctypes.c_double),
        ('spatial_weight', ctypes.c_double),
        ('spectral_weight', ctypes.c_double),
        ('sharded', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
    ]


class PbbssError(RuntimeError):
    pass


_lib = None
_lib_lock = threading.Lock()


def load():
    """dlopen the HIP library; raises if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PbbssError(
                f'{LIB_PATH} not found: build it with '
                '`make -C pb_bss_amd/csrc -j8` (or __graft_entry__.build()). '
                'There is no CPU fallback.')
        # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so).  It must be
        # mapped BEFORE this library is: the loader then resolves our libamdhip64.so.7 dependency
        # to that same copy.  Loaded the other way round, the process ends up with two HIP/HSA
        # runtimes and whichever initialises second reports "no ROCm-capable device".
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        lib.pbbss_version.restype = ctypes.c_int
        lib.pbbss_error_string.restype = ctypes.c_char_p
        lib.pbbss_error_string.argtypes = [i32]
        lib.pbbss_create.argtypes = [ctypes.POINTER(vp), i32]
        lib.pbbss_destroy.argtypes = [vp]
        lib.pbbss_set_timing.argtypes = [vp, i32]
        lib.pbbss_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        lib.pbbss_kernel_ms_lagged.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float)]
        lib.pbbss_set_phase_profile.argtypes = [vp, vp]
        lib.pbbss_set_split_tail.argtypes = [vp, i32]
        lib.pbbss_set_dhtv_team.argtypes = [vp, i32]
        lib.pbbss_set_dhtv_probe.argtypes = [vp, i32]
        lib.pbbss_split_error.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        lib.pbbss_split_reset.argtypes = [vp]
        lib.pbbss_set_spin_limit.argtypes = [vp, ctypes.c_uint]
        lib.pbbss_stft_num_frames.argtypes = [i64, i32, i32, i32, i32, i32]
        lib.pbbss_stft.argtypes = [vp, vp, i32, i64, i64, i32, i32, i32, vp, i32, i32, i32, i32, vp, vp]
        lib.pbbss_istft.argtypes = [vp, vp, i32, i64, i32, i32, i32, i32, vp, i32, vp, i64, vp]
        lib.pbbss_dhtv_calculate_mapping.argtypes = [vp, vp, i64, i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp]
        lib.pbbss_apply_mapping.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp]
        lib.pbbss_pa_pairwise_mapping.argtypes = [vp, vp, vp, i64, i32, i64, i32, vp, vp, i32, i32,
                                                  vp, vp, i64, i64, vp, vp]
        lib.pbbss_pa_compose_mapping.argtypes = [vp, vp, i64, i32, i64, vp]
        lib.pbbss_pa_mapping_from_scores.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp]
        lib.pbbss_cwmm_fit.argtypes = [
            vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, ctypes.POINTER(CwmmOpts), vp, vp,
            vp, vp, vp, vp, vp, vp, vp]
        lib.pbbss_normalize_observation.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp]
        lib.pbbss_cacgmm_fit.argtypes = [
            vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp,
            ctypes.POINTER(EmOpts), vp, vp, vp, vp, vp, vp, vp]
        lib.pbbss_cacgmm_fit_shared.argtypes = [
            vp, vp, i64, i32, i32, i32, i64, vp, vp, vp, vp, vp, vp,
            ctypes.POINTER(EmOpts), vp, vp, vp, vp, vp, vp, vp]
        lib.pbbss_cacgmm_predict.argtypes = [
            vp, vp, i64, i32, i32, i32, vp, vp, vp, i64, i64, i64, vp, i32, i32,
            dbl, vp, vp, vp, vp]
        lib.pbbss_cacg_m_step.argtypes = [
            vp, vp, i64, i32, i32, i32, vp, vp, i32, i32, i32, dbl, vp, vp, vp,
            vp, vp]
        lib.pbbss_heev_batched.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
        lib.pbbss_psd.argtypes = [vp, vp, i32, i64, i32, i32, i32, vp, i32, vp, vp]
        lib.pbbss_gev.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
        lib.pbbss_gev_general.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp]
        lib.pbbss_estimate_mixture_weight.argtypes = [vp, vp, vp, i64, i64, i32, i64, i32, i32, vp, vp]
        lib.pbbss_log_pdf_to_affiliation.argtypes = [vp, vp, i64, i32, i64, vp, i64, i64, i64, vp, dbl, vp, vp]
        lib.pbbss_log_pdf_to_affiliation_inline_pa.argtypes = [vp, vp, vp, i64, i32, i64, vp, i64, i64, i64, vp,
                                                               dbl, vp, vp, vp]
        lib.pbbss_comm_unique_id.argtypes = [vp]
        lib.pbbss_comm_create.argtypes = [vp, vp, i32, i32]
        lib.pbbss_comm_destroy.argtypes = [vp]
        lib.pbbss_comm_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        lib.pbbss_shard_bounds.argtypes = [i64, i32, i32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        lib.pbbss_allgather_masks.argtypes = [vp, vp, i32, i64, i64, i64, vp, vp]
        lib.pbbss_allgather_unpack.argtypes = [vp, vp, i32, i32, i64, i64, i64, vp, vp]
        lib.pbbss_solve.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp]
        lib.pbbss_mvdr_souden.argtypes = [vp, vp, vp, i64, i32, dbl, vp, vp, vp, vp, vp]
        lib.pbbss_mvdr.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
        lib.pbbss_wmwf.argtypes = [vp, vp, vp, i64, i32, dbl, i32, vp, vp, vp, vp, vp]
        lib.pbbss_ban.argtypes = [vp, vp, vp, i64, i32, vp, vp]
        lib.pbbss_apply_beamforming_vector.argtypes = [vp, vp, vp, i32, i64, i32, i32, vp, vp]
        lib.pbbss_apply_beamforming_vector_shared.argtypes = [vp, vp, vp, i32, i64, i64, i32, i32,
                                                              vp, vp]
        lib.pbbss_select_reference_channel.argtypes = [vp, vp, vp, vp, i64, i64, i32, i64, i64, dbl,
                                                       vp, vp, vp, vp]
        lib.pbbss_embed_log_pdf.argtypes = [vp, vp, i32, i64, i64, i32, i32, i32, vp, vp, vp, vp]
        lib.pbbss_embed_fit.argtypes = [vp, vp, i32, i64, i64, i32, i32, i32, i32, vp, dbl, dbl,
                                        vp, vp, vp]
        lib.pbbss_vmfmm_fit.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp,
                                        ctypes.POINTER(MixOpts), vp, vp, vp, vp, vp, vp]
        lib.pbbss_gmm_full_fit.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp,
                                           ctypes.POINTER(MixOpts), vp, vp, vp, vp, vp, vp, vp]
        lib.pbbss_gauss_full_fit.argtypes = [vp, vp, i32, i64, i64, i32, i32, vp, vp, vp, vp]
        lib.pbbss_gauss_full_log_pdf.argtypes = [vp, vp, i32, i64, i64, i32, i32, vp, vp, vp, vp, vp]
        lib.pbbss_gmm_fit.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp,
                                      ctypes.POINTER(MixOpts), vp, vp, vp, vp, vp, vp]
        lib.pbbss_joint_fit.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp,
                                        vp, vp, ctypes.POINTER(MixOpts), vp, vp, vp, vp, vp, vp,
                                        vp, vp]
        lib.pbbss_lcmv.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]
        lib.pbbss_phase_correction.argtypes = [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp]
        lib.pbbss_snr_postfilter.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp]
        lib.pbbss_reference_channel_terms.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp]
        lib.pbbss_rank_one_approximation.argtypes = [vp, vp, vp, i64, i32, vp, vp]
        lib.pbbss_matvec.argtypes = [vp, vp, vp, i64, i32, vp, vp]
        lib.pbbss_distortionless_normalization.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp]
        lib.pbbss_zero_degree_normalization.argtypes = [vp, vp, i64, i32, i32, vp, vp]
        lib.pbbss_condition_covariance.argtypes = [vp, vp, i64, i32, dbl, vp, vp]
        lib.pbbss_apply_online_beamforming_vector.argtypes = [vp, vp, vp, i32, i64, i32, i32, vp, vp]
        for name in EXPORTS:
            fn = getattr(lib, name)
            if name not in ('pbbss_error_string',):
                fn.restype = ctypes.c_int
        _lib = lib
        return lib


def check(rc, what=''):
    if rc != OK:
        msg = load().pbbss_error_string(rc).decode()
        if rc == ERR_UNSUPPORTED:
            raise NotImplementedError(f'{what}: {msg}')
        raise PbbssError(f'{what}: {msg} (code {rc})')


# ---- device handles -----------------------------------------------------------
_handles = {}
_handles_lock = threading.Lock()


def torch():
    import torch as _torch
    return _torch


def require_gpu():
    t = torch()
    if not t.cuda.is_available():
        raise PbbssError(
            'pb_bss_amd needs a ROCm GPU (torch.cuda.is_available() is False); '
            'there is no CPU fallback in the product path.')
    return t


def handle(device_index=None):
    """One C handle per (device, host thread)."""
    t = require_gpu()
    if device_index is None:
        device_index = t.cuda.current_device()
    key = (device_index, threading.get_ident())
    with _handles_lock:
        h = _handles.get(key)
        if h is None:
            lib = load()
            hp = ctypes.c_void_p()
            check(lib.pbbss_create(ctypes.byref(hp), device_index), 'pbbss_create')
            h = hp
            _handles[key] = h
        return h


def stream_ptr(device_index=None):
    t = torch()
    return ctypes.c_void_p(t.cuda.current_stream(device_index).cuda_stream)


def ptr(tensor):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if tensor is None:
        return None
    assert tensor.is_cuda and tensor.is_contiguous(), (tensor.device, tensor.stride())
    return ctypes.c_void_p(tensor.data_ptr())


# ---- host <-> device plumbing ---------------------------------------------------
def is_torch(x):
    return type(x).__module__.startswith('torch')


def to_device(x, dtype=None, device=None):
    """numpy array or torch tensor -> contiguous cuda tensor of `dtype`."""
    t = require_gpu()
    if device is None:
        device = t.device('cuda', t.cuda.current_device())
    if is_torch(x):
        out = x.to(device=device, dtype=dtype) if dtype is not None else x.to(device)
    else:
        arr = np.ascontiguousarray(x)
        out = t.from_numpy(arr).to(device)
        if dtype is not None and out.dtype != dtype:
            out = out.to(dtype)
    return out.contiguous()


def to_host(x):
    return x.detach().cpu().numpy()
