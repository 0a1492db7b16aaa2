"""
ctypes binding of libblhip.so (include/blhip.h).  Thin by design: structures, prototypes, error mapping.

There is no CPU fallback: if the shared library is missing, cannot be loaded, or no HIP device is visible,
:func:`load` / :class:`Context` raise :class:`~bayesloop_amd.exceptions.BackendError`.
"""
from __future__ import annotations

import ctypes as C
import os

from .exceptions import BackendError

LIB_NAME = 'libblhip.so'
ABI_VERSION = 8

OM_POISSON, OM_GAUSSIAN, OM_GAUSSIAN_MEAN, OM_TABLE = 1, 2, 3, 100
OM_BERNOULLI, OM_LAPLACE, OM_WHITE_NOISE, OM_AR1, OM_SCALED_AR1 = 4, 5, 6, 7, 8
OP_STATIC, OP_GRW, OP_CHANGEPOINT, OP_REGIMESWITCH, OP_INDEPENDENT, OP_BREAKPOINT, OP_NOTEQUAL = 0, 1, 2, 3, 4, 5, 6
OP_BIVARIATE, OP_BIVARIATE_ARG, OP_ALPHASTABLE, OP_ALPHASTABLE_ARG = 7, 8, 9, 10
OP_DETERMINISTIC, OP_DETERMINISTIC_ARG = 11, 12
FORWARD_ONLY, EVIDENCE_ONLY, KEEP_POSTERIOR, ACCUMULATE, RESUME, CARRY = 1, 2, 4, 8, 16, 32

c_double_p = C.POINTER(C.c_double)


class Op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('axis', C.c_int32), ('segment', C.c_int32), ('flags', C.c_int32)]


MAX_DIM = 4      # BLHIP_MAX_DIM


class Problem(C.Structure):
    _fields_ = [
        ('ndim', C.c_int32), ('obs_model', C.c_int32),
        ('n', C.c_int64 * MAX_DIM),
        ('marginal', c_double_p * MAX_DIM),
        ('lattice', C.c_double * MAX_DIM),
        ('T', C.c_int64),
        ('seg_len', C.c_int32), ('data_dim', C.c_int32),
        ('data', c_double_p), ('timestamps', c_double_p), ('prior', c_double_p), ('reset_prior', c_double_p),
        ('indep_prior', c_double_p), ('lik', c_double_p),
        ('n_ops', C.c_int32),
        ('ops', C.POINTER(Op)),
        ('resume_time', C.c_double), ('carry_slot', C.c_int32), ('reserved0', C.c_int32),
        ('backward_init', c_double_p),
        ('prior_token', C.c_uint64),
    ]


class Result(C.Structure):
    _fields_ = [
        ('log_evidence', c_double_p), ('local_evidence', c_double_p), ('posterior_mean', c_double_p),
        ('abort_step', C.POINTER(C.c_int64)), ('abort_phase', C.POINTER(C.c_int32)),
    ]


class Timing(C.Structure):
    _fields_ = [
        ('forward_ms', C.c_double), ('backward_ms', C.c_double), ('accumulate_ms', C.c_double), ('total_ms', C.c_double),
        ('forward_launches', C.c_int64), ('backward_launches', C.c_int64), ('accumulate_launches', C.c_int64),
        ('cells_per_launch', C.c_int64), ('batches', C.c_int64),
        ('fwd_kernel_variant', C.c_int32), ('bwd_kernel_variant', C.c_int32),
        ('fwd_hbm_bytes', C.c_double), ('bwd_hbm_bytes', C.c_double), ('fwd_flops', C.c_double), ('bwd_flops', C.c_double),
        ('resident_fallbacks', C.c_int32), ('resident_armed', C.c_int32),
        ('resident_fallback_reason', C.c_int32), ('peer_copy_path', C.c_int32),
        ('resident_probe', C.c_int32), ('xcd_order', C.c_int32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# name -> (restype, argtypes); the single source of truth checked against include/blhip.h by tests/test_abi.py
PROTOTYPES = {
    'blhip_abi_version': (C.c_int, []),
    'blhip_device_count': (C.c_int, []),
    'blhip_kernel_census': (C.c_int64, [C.c_char_p, C.c_int64]),
    'blhip_create': (C.c_void_p, [C.c_int]),
    'blhip_destroy': (None, [C.c_void_p]),
    'blhip_last_error': (C.c_char_p, [C.c_void_p]),
    'blhip_device_name': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    'blhip_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    'blhip_synchronize': (C.c_int, [C.c_void_p]),
    'blhip_fit': (C.c_int, [C.c_void_p, C.POINTER(Problem), C.c_int64, c_double_p, c_double_p, C.c_uint32,
                            C.POINTER(Result)]),
    'blhip_last_timing': (C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    'blhip_bandwidth_probe': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, c_double_p]),
    'blhip_host_alloc': (C.c_void_p, [C.c_size_t]),
    'blhip_host_free': (None, [C.c_void_p]),
    'blhip_posterior_read': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, c_double_p]),
    'blhip_posterior_devptr': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'blhip_posterior_release': (C.c_int, [C.c_void_p]),
    'blhip_posterior_marginal': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, c_double_p]),
    'blhip_posterior_time_average': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_double_p]),
    'blhip_accum_begin': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    'blhip_accum_state': (C.c_int, [C.c_void_p, c_double_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    'blhip_accum_rescale': (C.c_int, [C.c_void_p, C.c_double]),
    'blhip_accum_fold_host': (C.c_int, [C.c_void_p, c_double_p, C.c_double]),
    'blhip_accum_finalize': (C.c_int, [C.c_void_p, C.POINTER(Problem), c_double_p]),
    'blhip_accum_read': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_double_p]),
    'blhip_accum_end': (C.c_int, [C.c_void_p]),
    'blhip_accum_row_stats': (C.c_int, [C.c_void_p, C.POINTER(Problem), c_double_p]),
    'blhip_comm_unique_id': (C.c_int, [C.c_char_p]),
    'blhip_comm_init': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    'blhip_comm_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'blhip_comm_allgather': (C.c_int, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    'blhip_comm_allreduce': (C.c_int, [C.c_void_p, c_double_p, C.c_int64, C.c_int]),
    'blhip_comm_reduce_accum': (C.c_int, [C.c_void_p, C.c_int]),
    'blhip_comm_timing': (C.c_int, [C.c_void_p, c_double_p]),
    'blhip_comm_destroy': (C.c_int, [C.c_void_p]),
    'blhip_accum_peer_reduce': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64]),
    'blhip_accum_peer_gather': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'blhip_carry_mix': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_double_p, C.c_int]),
    'blhip_carry_read': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_double_p]),
    'blhip_carry_write': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, c_double_p]),
    'blhip_host_unlag': (C.c_int, [C.c_int, c_double_p, C.c_int64, C.c_int, C.POINTER(C.c_ubyte), c_double_p]),
    'blhip_carry_release': (C.c_int, [C.c_void_p, C.c_int]),
}

_lib = None


def library_path():
    env = os.environ.get('BLHIP_LIBRARY')
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def load():
    """Loads libblhip.so (built in-tree by ``__graft_entry__.build()`` / ``bayesloop_amd/csrc/build.py``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise BackendError('%s not found: build it with `python -m bayesloop_amd.csrc.build` '
                           '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise BackendError('cannot load %s: %s' % (path, e)) from e
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BackendError('%s does not export %s' % (path, name)) from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.blhip_abi_version() != ABI_VERSION:
        raise BackendError('%s has ABI version %d, expected %d' % (path, lib.blhip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def dptr(a):
    """float64 C-contiguous numpy array -> double* (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(c_double_p)
