"""ctypes binding of the C ABI in include/surfh_amd.h (libsurfh_amd.so, built in-tree by
``__graft_entry__.build()``).  There is no CPU fallback: if the library is missing or a
call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsurfh_amd.so")

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
# int cb(void *user, int32_t it, const double *grad_norm, const float *x)   (include/surfh_amd.h:surfh_cg_callback)
CG_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, c_double_p, c_float_p)
c_uint8_p = C.POINTER(C.c_uint8)


class ChannelDesc(C.Structure):
    _fields_ = [
        ("wslice_start", C.c_int32), ("wslice_stop", C.c_int32),
        ("n_pointings", C.c_int32), ("n_slit", C.c_int32), ("n_lambda_out", C.c_int32),
        ("n_alpha_out", C.c_int32), ("srf", C.c_int32), ("na", C.c_int32), ("nb", C.c_int32),
        ("alpha0", C.c_int32), ("n_alpha_slit", C.c_int32), ("n_beta_slit", C.c_int32),
        ("slit_beta0", c_int32_p), ("slit_weights", c_double_p),
        ("grid_i0", c_int32_p), ("grid_i1", c_int32_p), ("grid_y0", c_double_p), ("grid_y1", c_double_p),
        ("wpsf", c_double_p),
        ("gt_i0", c_int32_p), ("gt_i1", c_int32_p), ("gt_y0", c_double_p), ("gt_y1", c_double_p),
        ("gt_inside", c_uint8_p),
        ("box_len", C.c_int32), ("box_shift", C.c_int32),
    ]


class Config(C.Structure):
    _fields_ = [
        ("n_alpha", C.c_int32), ("n_beta", C.c_int32), ("n_lambda", C.c_int32), ("n_templates", C.c_int32),
        ("templates", c_double_p), ("sotf", c_double_p),
        ("n_channels", C.c_int32), ("channels", C.POINTER(ChannelDesc)),
        ("device", C.c_int32), ("stream", C.c_void_p), ("split_k_forward", C.c_int32), ("verify", C.c_int32),
        ("exact", C.c_int32),
    ]


EXPORTS = [
    "surfh_last_error", "surfh_version", "surfh_plan_create", "surfh_plan_destroy", "surfh_isize", "surfh_osize",
    "surfh_stream", "surfh_forward", "surfh_adjoint", "surfh_adjoint_ref", "surfh_fwadj", "surfh_forward_dev",
    "surfh_adjoint_dev", "surfh_adjoint_ref_dev", "surfh_fwadj_dev", "surfh_wct_forward", "surfh_wct_adjoint",
    "surfh_wct_fwadj", "surfh_wct_expsol", "surfh_tst_create", "surfh_tst_destroy", "surfh_tst_forward",
    "surfh_tst_adjoint", "surfh_tst_fwadj", "surfh_tst_last_error", "surfh_cg", "surfh_cg_cb", "surfh_mmmg", "surfh_cg_planes", "surfh_mmmg_planes", "surfh_cg_planes_cb", "surfh_mmmg_planes_cb", "surfh_cg_planes_begin_dev", "surfh_cg_planes_step_dev", "surfh_cg_planes_rr", "surfh_maps_to_cube", "surfh_cube_to_maps", "surfh_normal_dev",
    "surfh_prior_add_dev", "surfh_spec_supported", "surfh_spec_size", "surfh_to_spec_dev", "surfh_from_spec_dev", "surfh_forward_spec_dev",
    "surfh_adjoint_spec_dev", "surfh_normal_spec_dev", "surfh_prior_spec_add_dev", "surfh_set_prior", "surfh_dot_dev", "surfh_cg_step_dev", "surfh_cg_dir_dev", "surfh_cg_iter_dev", "surfh_residual_dev",
    "surfh_cg_begin_dev", "surfh_cg_iter_nosync_dev", "surfh_cg_xupdate_nosync_dev", "surfh_cg_refresh_nosync_dev", "surfh_cg_trace",
    "surfh_profile_enable", "surfh_profile_filter", "surfh_profile_count", "surfh_profile_get", "surfh_profile_reset", "surfh_debug_copy",
    "surfh_debug_dims", "surfh_gemm_selftest", "surfh_gemm_selftest_ksteps", "surfh_klist_classify",
]

_lib = None


def load():
    """Load libsurfh_amd.so and declare the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP path has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.surfh_last_error.restype = C.c_char_p
    L.surfh_version.restype = C.c_int
    L.surfh_plan_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.surfh_plan_destroy.argtypes = [vp]
    L.surfh_isize.argtypes = [vp]; L.surfh_isize.restype = C.c_int64
    L.surfh_osize.argtypes = [vp]; L.surfh_osize.restype = C.c_int64
    L.surfh_stream.argtypes = [vp]; L.surfh_stream.restype = vp
    for n in ("surfh_forward", "surfh_adjoint", "surfh_adjoint_ref", "surfh_fwadj", "surfh_wct_forward", "surfh_wct_adjoint",
              "surfh_wct_fwadj"):
        getattr(L, n).argtypes = [vp, c_float_p, c_float_p]
    for n in ("surfh_forward_dev", "surfh_adjoint_dev", "surfh_adjoint_ref_dev", "surfh_fwadj_dev"):
        getattr(L, n).argtypes = [vp, vp, vp]
    L.surfh_cg.argtypes = [vp, c_float_p, C.c_double, C.c_double, c_float_p, C.c_int32, C.c_double, C.c_int32,
                           c_float_p, c_double_p, c_int32_p]
    L.surfh_tst_create.argtypes = [C.c_int32] * 4 + [c_double_p, c_int32_p, C.c_int64, c_float_p, C.c_int32, C.POINTER(vp)]
    L.surfh_tst_destroy.argtypes = [vp]
    for n in ("surfh_tst_forward", "surfh_tst_adjoint", "surfh_tst_fwadj"):
        getattr(L, n).argtypes = [vp, c_float_p, c_float_p]
    L.surfh_tst_last_error.restype = C.c_char_p
    L.surfh_wct_expsol.argtypes = [vp, c_float_p, c_double_p, c_double_p, c_float_p]
    L.surfh_cg_planes.argtypes = [vp, c_float_p, C.c_double, C.c_double, c_float_p, C.c_int32, C.c_double, C.c_int32,
                                  c_float_p, c_double_p, c_int32_p]
    L.surfh_cg_cb.argtypes = [vp, c_float_p, C.c_double, C.c_double, c_float_p, C.c_int32, C.c_double, C.c_int32,
                              c_float_p, c_double_p, c_int32_p, CG_CALLBACK, vp]
    L.surfh_mmmg.argtypes = L.surfh_cg_cb.argtypes
    L.surfh_mmmg_planes.argtypes = L.surfh_cg_planes.argtypes
    L.surfh_cg_planes_begin_dev.argtypes = [vp, vp, C.c_double, C.c_double, vp]
    L.surfh_cg_planes_step_dev.argtypes = [vp, C.c_int32, C.c_int32]
    L.surfh_cg_planes_rr.argtypes = [vp, c_double_p]
    L.surfh_cg_planes_cb.argtypes = L.surfh_cg_cb.argtypes
    L.surfh_mmmg_planes_cb.argtypes = L.surfh_cg_cb.argtypes
    L.surfh_maps_to_cube.argtypes = [vp, c_double_p, C.c_int32, C.c_int32, c_float_p, c_float_p]
    L.surfh_cube_to_maps.argtypes = [vp, c_double_p, C.c_int32, C.c_int32, c_float_p, c_float_p]
    L.surfh_normal_dev.argtypes = [vp, vp, vp, C.c_double]
    L.surfh_prior_add_dev.argtypes = [vp, vp, vp, C.c_double]
    L.surfh_spec_supported.argtypes = [vp]
    L.surfh_spec_size.argtypes = [vp]
    L.surfh_spec_size.restype = C.c_int64
    L.surfh_to_spec_dev.argtypes = [vp, vp, vp]
    L.surfh_from_spec_dev.argtypes = [vp, vp, vp]
    L.surfh_forward_spec_dev.argtypes = [vp, vp, vp]
    L.surfh_adjoint_spec_dev.argtypes = [vp, vp, vp, C.c_double, vp, C.c_double]
    L.surfh_normal_spec_dev.argtypes = [vp, vp, vp, C.c_double, C.c_double]
    L.surfh_prior_spec_add_dev.argtypes = [vp, vp, vp, C.c_double]
    L.surfh_set_prior.argtypes = [vp, C.c_int32]
    L.surfh_dot_dev.argtypes = [vp, vp, vp, C.c_int64, c_double_p]
    L.surfh_cg_step_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_double, c_double_p]
    L.surfh_cg_dir_dev.argtypes = [vp, vp, vp, C.c_int64, C.c_double]
    L.surfh_cg_iter_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int64, C.c_double, c_double_p]
    L.surfh_residual_dev.argtypes = [vp, vp, vp, vp, C.c_int64]
    L.surfh_cg_begin_dev.argtypes = [vp, vp, C.c_int64]
    L.surfh_cg_iter_nosync_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int64]
    L.surfh_cg_xupdate_nosync_dev.argtypes = [vp, vp, vp, vp, C.c_int64]
    L.surfh_cg_refresh_nosync_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int64]
    L.surfh_cg_trace.argtypes = [vp, c_double_p, C.c_int32]; L.surfh_cg_trace.restype = C.c_int32
    L.surfh_profile_enable.argtypes = [vp, C.c_int32]
    L.surfh_profile_filter.argtypes = [vp, C.c_char_p]
    L.surfh_profile_count.argtypes = [vp]; L.surfh_profile_count.restype = C.c_int32
    L.surfh_profile_get.argtypes = [vp, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), c_double_p]
    L.surfh_profile_reset.argtypes = [vp]
    L.surfh_debug_copy.argtypes = [vp, C.c_char_p, c_float_p, C.c_int64]; L.surfh_debug_copy.restype = C.c_int64
    L.surfh_debug_dims.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.surfh_gemm_selftest.argtypes = [C.c_int32] * 5 + [c_float_p, c_float_p, c_float_p]
    L.surfh_gemm_selftest_ksteps.argtypes = [C.POINTER(C.c_int64)]
    L.surfh_klist_classify.argtypes = [c_float_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int64]
    L.surfh_klist_classify.restype = C.c_int32
    _lib = L
    return L


def check(rc: int, exc=RuntimeError):
    if rc != 0:
        raise exc(load().surfh_last_error().decode("utf-8", "replace"))


def fptr(a: np.ndarray):
    return a.ctypes.data_as(c_float_p)


def dptr(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def iptr(a: np.ndarray):
    return a.ctypes.data_as(c_int32_p)


def u8ptr(a: np.ndarray):
    return a.ctypes.data_as(c_uint8_p)
