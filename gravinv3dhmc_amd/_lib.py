"""ctypes binding of libgravhmc.so (the C-ABI of include/gravhmc.h).

There is no CPU implementation behind this module: if the shared library is missing, or
the process has no usable HIP device, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
#: GRAVHMC_LIB: another build of the library (A/B measurements of one kernel change on one box)
LIB_PATH = os.environ.get("GRAVHMC_LIB") or os.path.join(_HERE, "libgravhmc.so")

GH_OK, GH_ERR_ARG, GH_ERR_HIP, GH_ERR_NOMEM, GH_ERR_OVERFLOW, GH_ERR_UNSUPPORTED, GH_ERR_COMM = \
    0, -1, -2, -3, -4, -5, -6
CELL_PRISM, CELL_TESSEROID = 0, 1
REG_KINDS = {"Damping": 0, "Smoothness": 1, "MS": 2, "TV": 3}

_dp = C.POINTER(C.c_double)
_ctx = C.c_void_p
_i64 = C.c_int64

#: every symbol include/gravhmc.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "gh_create": (C.c_int, [C.POINTER(_ctx), C.c_int, _i64, _i64]),
    "gh_destroy": (None, [_ctx]),
    "gh_last_error": (C.c_char_p, [_ctx]),
    "gh_device_info": (C.c_int, [_ctx, C.c_char_p, C.POINTER(C.c_int), C.POINTER(_i64)]),
    "gh_synchronize": (C.c_int, [_ctx]),
    "gh_set_obs": (C.c_int, [_ctx, _dp, _dp, _dp]),
    "gh_set_cells": (C.c_int, [_ctx, _dp, C.c_int, C.c_double]),
    "gh_set_matrix_free": (C.c_int, [_ctx, C.c_int]),
    "gh_set_matrix_free_exact": (C.c_int, [_ctx, C.c_int]),
    "gh_batch_fused_stats": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int)]),
    "gh_matrix_free_team_stats": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int)]),
    "gh_batch_resident_stats": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "gh_set_shift_invariant": (C.c_int, [_ctx, C.c_int]),
    "gh_shift_invariant_info": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int64)]),
    "gh_pinned_alloc": (C.c_int, [_ctx, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gh_pinned_free": (C.c_int, [_ctx, C.c_void_p]),
    "gh_batch_staging_stats": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gh_shift_invariant_resident_stats": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "gh_shift_invariant_harmonic": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                              C.POINTER(C.c_int)]),
    "gh_matrix_free_stats": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                                       C.POINTER(_i64), C.POINTER(_i64)]),
    "gh_build_G": (C.c_int, [_ctx]),
    "gh_kernel_stats": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64)]),
    "gh_upload_G": (C.c_int, [_ctx, _dp, _i64, C.c_int]),
    "gh_download_G": (C.c_int, [_ctx, _dp, _i64]),
    "gh_weight": (C.c_int, [_ctx, C.c_double, _dp]),
    "gh_set_data": (C.c_int, [_ctx, _dp, _dp]),
    "gh_set_reg": (C.c_int, [_ctx, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int), _dp]),
    "gh_forward": (C.c_int, [_ctx, _dp, _dp]),
    "gh_adjoint": (C.c_int, [_ctx, _dp, _dp]),
    "gh_misfit_and_grad": (C.c_int, [_ctx, _dp, _dp, _dp, _dp]),
    "gh_reg_eval": (C.c_int, [_ctx, C.c_int, C.c_double, C.POINTER(C.c_int), C.c_int, _dp, _dp,
                              C.POINTER(C.c_double), _dp]),
    "gh_compress_wavelet": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_int), C.c_double, C.c_int,
                                      C.POINTER(_i64), C.POINTER(_i64)]),
    "gh_download_csr": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(C.c_int32), _dp]),
    "gh_model_coeffs": (C.c_int, [_ctx, _dp, _dp]),
    "gh_forward_wavelet": (C.c_int, [_ctx, _dp, _dp]),
    "gh_chain_init": (C.c_int, [_ctx, _dp, _dp, _dp]),
    "gh_chain_trajectory": (C.c_int, [_ctx, _dp, C.c_double, C.c_int, C.c_double,
                                      C.POINTER(C.c_int), _dp]),
    "gh_chain_prefetch_momentum": (C.c_int, [_ctx, _dp]),
    "gh_chain_stats": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64)]),
    "gh_chain_resident_stats": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64)]),
    "gh_team_sweep_stats": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(_i64), C.POINTER(C.c_int),
                                      C.POINTER(_i64)]),
    "gh_chain_run": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_int), _dp, _dp, C.c_double, _dp, _i64, _i64,
                               C.POINTER(C.c_int), _dp, _dp, C.POINTER(C.c_int)]),
    "gh_chain_get_x": (C.c_int, [_ctx, _dp]),
    "gh_chain_get_dsyn": (C.c_int, [_ctx, _dp]),
    "gh_batch_init": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp]),
    "gh_batch_trajectory": (C.c_int, [_ctx, _dp, C.c_double, C.POINTER(C.c_int), _dp,
                                      C.POINTER(C.c_int), _dp]),
    "gh_batch_run": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_int), C.POINTER(_dp), _dp, C.c_double,
                               C.POINTER(C.c_int), _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gh_batch_get_x": (C.c_int, [_ctx, C.c_int, _dp]),
    "gh_posterior_window": (C.c_int, [_ctx, C.c_int]),
    "gh_posterior_add": (C.c_int, [_ctx]),
    "gh_posterior_read": (C.c_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64), _dp, _dp]),
    "gh_leapfrog": (C.c_int, [_ctx, _dp, _dp, C.c_double, C.c_int, _dp, _dp, C.c_double,
                              C.POINTER(C.c_int), _dp, _dp]),
    "gh_shard_unique_id": (C.c_int, [C.c_void_p]),
    "gh_shard_init": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, _i64, _i64]),
    "gh_shard_init_callback": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i64, _i64]),
    "gh_shard_init_rows": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, _i64, _i64]),
    "gh_shard_init_rows_callback": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i64, _i64]),
    "gh_shard_allreduce": (C.c_int, [_ctx, _dp, _i64]),
    "gh_rng_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32]),
    "gh_rng_destroy": (None, [C.c_void_p]),
    "gh_rng_set_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "gh_host_cores": (C.c_int, []),
    "gh_rng_set_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_double]),
    "gh_rng_get_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_double)]),
    "gh_rng_draw_trajectories": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _i64, C.c_double,
                                           C.POINTER(C.c_int), _dp, _dp]),
    "gh_format_row_fixed8": (C.c_int64, [_dp, _i64, C.c_char_p, _i64]),
    "gh_measure_stream_read": (C.c_int, [_ctx, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gh_profile_enable": (C.c_int, [_ctx, C.c_int]),
    "gh_profile_read": (C.c_int, [_ctx, C.POINTER(C.c_double), C.POINTER(_i64),
                                  C.POINTER(_i64)]),
}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, _i64)

_LIB = None


class GravHmcError(RuntimeError):
    pass


def load():
    """dlopen libgravhmc.so and bind every prototype; raises if the library is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GravHmcError(
                "libgravhmc.so is not built (%s). Run `python -m gravinv3dhmc_amd.build` "
                "(needs hipcc); there is no CPU fallback." % LIB_PATH)
        if "GRAVHMC_BATCH_TEAM" not in os.environ and not os.environ.get("GRAVHMC_LIB"):
            # batch_team_kernel covers its inline-assembly loads with hand-counted waits: the library keeps
            # that form only when the code THIS build's compiler generated for it passed the scan
            # (isa_check.py, run by build()); otherwise the two-pass batch kernels take its place
            from . import isa_check
            if not isa_check.team_form_cleared():
                import sys
                sys.stderr.write("gravinv3dhmc_amd: the generated code of batch_team_kernel was not cleared by the "
                                 "build's scan (%s): using the two-pass batch kernels\n" % isa_check.STAMP)
                os.environ["GRAVHMC_BATCH_TEAM"] = "0"
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def check(rc, ctx=None):
    if rc == GH_OK:
        return
    msg = load().gh_last_error(ctx)
    msg = msg.decode("utf-8", "replace") if msg else "libgravhmc error %d" % rc
    if rc == GH_ERR_ARG:
        raise ValueError(msg)
    if rc == GH_ERR_OVERFLOW:
        raise OverflowError(msg)
    if rc == GH_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == GH_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise GravHmcError(msg)
