"""ctypes binding of librls_mi355x.so (C ABI: include/rls_mi355x.h).

There is NO fallback: if the shared library is missing, or a call returns a non-zero status, this
module raises.  The product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librls_mi355x.so")

F32, C32 = 0, 1
F64, C64 = 2, 3   # the rls_*_d entry points only (the L1 protocol with double scalars)
OP_N, OP_T, OP_C = 0, 1, 2
REG_NONE, REG_L1, REG_L2, REG_L21, REG_TV = 0, 1, 2, 3, 4
PROJ_NONE, PROJ_REAL, PROJ_POSITIVE = 0, 1, 2


class RLSError(RuntimeError):
    pass


class CgnrStatus(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("done", C.c_int32), ("alpha_re", C.c_float), ("alpha_im", C.c_float),
                ("beta_re", C.c_float), ("beta_im", C.c_float), ("zeta", C.c_float), ("residual", C.c_float),
                ("z0", C.c_float), ("fallbacks", C.c_int32)]


class FistaStatus(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("done", C.c_int32), ("theta", C.c_float), ("theta_old", C.c_float),
                ("rel_res_norm", C.c_float), ("residual", C.c_float), ("norm_x0", C.c_float), ("fallbacks", C.c_int32)]


class CgStatus(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("residual", C.c_float), ("tol", C.c_float), ("fallbacks", C.c_int32)]


class AdmmParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("xold", C.c_void_p), ("beta", C.c_void_p), ("beta_y", C.c_void_p),
                ("z0", C.c_void_p), ("z1", C.c_void_p), ("u", C.c_void_p), ("rho", C.c_float),
                ("sigma_abs", C.c_float), ("rel_tol", C.c_float), ("iterations", C.c_int32),
                ("iterations_cg", C.c_int32), ("tol_inner", C.c_float), ("reg_kind", C.c_int32),
                ("prox_lambda", C.c_float), ("proj_kind", C.c_int32), ("tv_ndims", C.c_int32), ("tv_ntv", C.c_int32),
                ("tv_iterations", C.c_int32), ("tv_dims", C.c_int32 * 4), ("tv_shape", C.c_int64 * 4)]


class AdmmStatus(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("done", C.c_int32), ("rk", C.c_float), ("sk", C.c_float),
                ("eps_pri", C.c_float), ("eps_dua", C.c_float), ("delta", C.c_float), ("cg_iterations", C.c_int32),
                ("fallbacks", C.c_int32)]


_vp, _i32, _i64, _f, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_pvp = C.POINTER(C.c_void_p)
_pf = C.POINTER(C.c_float)
_d = C.c_double
_pd = C.POINTER(C.c_double)
_pi64 = C.POINTER(C.c_int64)
_pi32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes).  Must list every symbol the header declares (tests/test_abi.py checks).
PROTOTYPES = {
    "rls_abi_version": (_i32, []),
    "rls_ctx_create": (_i32, [_i32, _pvp]),
    "rls_ctx_create_on_stream": (_i32, [_i32, _vp, _pvp]),
    "rls_ctx_destroy": (_i32, [_vp]),
    "rls_ctx_sync": (_i32, [_vp]),
    "rls_ctx_stream": (_vp, [_vp]),
    "rls_last_error_string": (C.c_char_p, [_vp]),
    "rls_device_count": (_i32, [_pi32]),
    "rls_tune_set": (_i32, [_vp, C.c_char_p, _i32]),
    "rls_malloc": (_i32, [_vp, _sz, _pvp]),
    "rls_free": (_i32, [_vp, _vp]),
    "rls_memcpy_h2d": (_i32, [_vp, _vp, _vp, _sz]),
    "rls_memcpy_d2h": (_i32, [_vp, _vp, _vp, _sz]),
    "rls_memcpy_d2d": (_i32, [_vp, _vp, _vp, _sz]),
    "rls_fill": (_i32, [_vp, _i32, _i64, _vp, _f, _f]),
    "rls_timer_start": (_i32, [_vp]),
    "rls_timer_stop_ms": (_i32, [_vp, _pf]),
    "rls_gemv": (_i32, [_vp, _i32, _i32, _i64, _i64, _f, _f, _vp, _i64, _vp, _f, _f, _vp]),
    "rls_nrm2": (_i32, [_vp, _i32, _i64, _vp, _pf]),
    "rls_nrm2_dev": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "rls_dotc": (_i32, [_vp, _i32, _i64, _vp, _vp, _pf]),
    "rls_dotc_dev": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "rls_asum": (_i32, [_vp, _i32, _i64, _vp, _pf]),
    "rls_scal": (_i32, [_vp, _i32, _i64, _f, _f, _vp]),
    "rls_axpy": (_i32, [_vp, _i32, _i64, _f, _f, _vp, _vp]),
    "rls_axpby": (_i32, [_vp, _i32, _i64, _f, _f, _vp, _f, _f, _vp]),
    "rls_lincomb": (_i32, [_vp, _i32, _i64, _f, _f, _vp, _f, _f, _vp, _vp]),
    # ---- Float64 / ComplexF64: the L1 protocol with double scalars (csrc/f64.hip) ----
    "rls_fill_d": (_i32, [_vp, _i32, _i64, _vp, _d, _d]),
    "rls_scal_d": (_i32, [_vp, _i32, _i64, _d, _d, _vp]),
    "rls_axpy_d": (_i32, [_vp, _i32, _i64, _d, _d, _vp, _vp]),
    "rls_lincomb_d": (_i32, [_vp, _i32, _i64, _d, _d, _vp, _d, _d, _vp, _vp]),
    "rls_nrm2_d": (_i32, [_vp, _i32, _i64, _vp, _pd]),
    "rls_asum_d": (_i32, [_vp, _i32, _i64, _vp, _pd]),
    "rls_dotc_d": (_i32, [_vp, _i32, _i64, _vp, _vp, _pd]),
    "rls_gemv_d": (_i32, [_vp, _i32, _i32, _i64, _i64, _d, _d, _vp, _i64, _vp, _d, _d, _vp]),
    "rls_prox_l1_d": (_i32, [_vp, _i32, _i64, _vp, _d]),
    "rls_prox_l2_d": (_i32, [_vp, _i32, _i64, _vp, _d]),
    "rls_prox_l21_d": (_i32, [_vp, _i32, _i64, _i64, _vp, _d]),
    "rls_prox_positive_d": (_i32, [_vp, _i32, _i64, _vp]),
    "rls_prox_real_d": (_i32, [_vp, _i32, _i64, _vp]),
    "rls_prox_tv_fgp_d": (_i32, [_vp, _i32, _i32, _pi64, _i32, _pi32, _vp, _d, _i32]),
    "rls_transpose_d": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _vp, _i64]),
    "rls_rownorm2_d": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _vp]),
    "rls_scale_rows_d": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp, _i64, _vp, _i64]),
    "rls_kaczmarz_sweep_d": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _d, _i32]),
    "rls_prox_l1": (_i32, [_vp, _i32, _i64, _vp, _f]),
    "rls_prox_l2": (_i32, [_vp, _i32, _i64, _vp, _f]),
    "rls_prox_l21": (_i32, [_vp, _i32, _i64, _i64, _vp, _f]),
    "rls_prox_positive": (_i32, [_vp, _i32, _i64, _vp]),
    "rls_prox_real": (_i32, [_vp, _i32, _i64, _vp]),
    "rls_norm_l21": (_i32, [_vp, _i32, _i64, _i64, _vp, _f, _pf]),
    "rls_tv_grad_len": (_i64, [_i32, _pi64, _i32, _pi32]),
    "rls_tv_grad": (_i32, [_vp, _i32, _i32, _pi64, _i32, _pi32, _vp, _vp, _f, _f]),
    "rls_tv_grad_t": (_i32, [_vp, _i32, _i32, _pi64, _i32, _pi32, _vp, _vp, _f, _f]),
    "rls_tv_restrict": (_i32, [_vp, _i32, _i64, _vp]),
    "rls_tv_lincomb": (_i32, [_vp, _i32, _i64, _vp, _f, _vp, _f, _vp]),
    "rls_prox_tv_workspace_bytes": (_sz, [_i32, _i32, _pi64, _i32, _pi32]),
    "rls_prox_tv_fgp": (_i32, [_vp, _i32, _i32, _pi64, _i32, _pi32, _vp, _f, _i32, _vp, _sz]),
    "rls_operator_create": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _pvp]),
    "rls_operator_set_gram": (_i32, [_vp, _vp, _i64]),
    "rls_operator_destroy": (_i32, [_vp]),
    "rls_operator_mul": (_i32, [_vp, _vp, _vp]),
    "rls_operator_mul_adj": (_i32, [_vp, _vp, _vp]),
    "rls_operator_mul_normal": (_i32, [_vp, _vp, _vp]),
    "rls_gram": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _vp, _i64]),
    "rls_prox_nuclear": (_i32, [_vp, _i32, _i64, _i64, _vp, _f]),
    "rls_prox_llr": (_i32, [_vp, _i32, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i64, _vp, _f]),
    "rls_optista_update": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i32, _f, _f, _f, _f, _f, _f, _pf]),
    "rls_pogm_update": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i32, _f, _i32, _i32,
                               _f, _pf]),
    "rls_optista_update_async": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i32, _f, _f, _f, _f, _f, _f,
                                        _f, _f, _vp]),
    "rls_pogm_update_async": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i32, _f, _i32,
                                     _f, _f, _vp]),
    "rls_pgm_create": (_i32, [_vp, C.POINTER(_vp)]),
    "rls_pgm_destroy": (_i32, [_vp]),
    "rls_pgm_step_resident": (_i32, [_vp, _i32, _i32, _i32, _pf, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _f, _vp]),
    "rls_pgm_lost": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "rls_pogm_update_auto": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i32, _i32, _i32, _f, _f, _vp]),
    "rls_pogm_step_resident_restart": (_i32, [_vp, _i32, _i32, _f, _f, _f, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f, _f, _vp]),
    "rls_operator_mul_normal_skip": (_i32, [_vp, _vp, _vp, _vp]),
    "rls_transpose": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _vp, _i64]),
    "rls_kaczmarz_sweep": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32,
                                  C.c_float, _i32]),
    "rls_rownorm2": (_i32, [_vp, _i32, _i64, _i64, _vp, _i64, _vp]),
    "rls_scale_rows": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp, _i64, _vp, _i64]),
    "rls_cgnr_create": (_i32, [_vp, _vp, _vp, _vp, _vp, _pvp]),
    "rls_cgnr_destroy": (_i32, [_vp]),
    "rls_cgnr_init": (_i32, [_vp, _vp, _f, _f, _i32]),
    "rls_cgnr_step": (_i32, [_vp, _i32]),
    "rls_cgnr_get_status": (_i32, [_vp, C.POINTER(CgnrStatus)]),
    "rls_cgnr_step_status": (_i32, [_vp, _i32, C.POINTER(CgnrStatus)]),
    "rls_cgnr_step_group": (_i32, [C.POINTER(C.c_void_p), _i32, _i32]),
    "rls_cgnr_get_status_group": (_i32, [C.POINTER(C.c_void_p), _i32, C.POINTER(CgnrStatus)]),
    "rls_cgnr_init_step_group": (_i32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i32, _f, _f, _i32, _i32]),
    "rls_cgnr_solve_queue": (_i32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i32, _f, _f, _i32, _vp]),
    "rls_cgnr_solve_queue_host": (_i32, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i32, _f, _f, _i32, _vp]),
    "rls_cgnr_create_batched": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _pvp]),
    "rls_cgnr_init_batched": (_i32, [_vp, _vp, _i64, _f, _f, _i32]),
    "rls_cgnr_get_status_batched": (_i32, [_vp, C.POINTER(CgnrStatus)]),
    "rls_cgnr_step_profiled": (_i32, [_vp, _i32, _pf, _pf]),
    "rls_cgnr_path": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "rls_cgnr_init_local_a": (_i32, [_vp, _vp, _f, _f, _i32]),
    "rls_cgnr_init_local_b": (_i32, [_vp]),
    "rls_cgnr_step_local_a": (_i32, [_vp]),
    "rls_cgnr_step_local_b": (_i32, [_vp]),
    "rls_fista_create": (_i32, [_vp, _vp, _vp, _vp, _vp, _pvp]),
    "rls_fista_destroy": (_i32, [_vp]),
    "rls_fista_set_reg": (_i32, [_vp, _i32, _f, _i64, _i32]),
    "rls_fista_set_reg_tv": (_i32, [_vp, _f, _i32, C.POINTER(C.c_int64), _i32, C.POINTER(C.c_int32), _i32, _i32]),
    "rls_fista_init": (_i32, [_vp, _vp, _f, _f, _f, _i32, _i32]),
    "rls_fista_set_start": (_i32, [_vp, _vp, C.c_int64]),
    "rls_fista_step": (_i32, [_vp, _i32]),
    "rls_fista_path": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "rls_fista_get_status": (_i32, [_vp, C.POINTER(FistaStatus)]),
    "rls_fista_step_status": (_i32, [_vp, _i32, C.POINTER(FistaStatus)]),
    "rls_fista_solution": (_i32, [_vp, _pvp]),
    "rls_cg_create": (_i32, [_vp, _vp, _vp, _vp, _pvp]),
    "rls_cg_create_batched": (_i32, [_vp, _i32, _vp, _vp, _vp, C.c_int64, _pvp]),
    "rls_cg_destroy": (_i32, [_vp]),
    "rls_cg_solve": (_i32, [_vp, _vp, _vp, _f, _i32, _f]),
    "rls_cg_get_status": (_i32, [_vp, C.POINTER(CgStatus)]),
    "rls_cg_path": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "rls_admm_pre": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i32]),
    "rls_admm_post": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _pf]),
    "rls_gather": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "rls_scatter": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "rls_stats": (_i32, [_vp, _i32, _i64, _vp, C.POINTER(C.c_double)]),
    "rls_shift_scale": (_i32, [_vp, _i64, _vp, _f, _f, _i32]),
    "rls_clamp": (_i32, [_vp, _i64, _vp, _f, _f]),
    "rls_restore_outside": (_i32, [_vp, _i64, _vp, _vp, _f, _f]),
    "rls_complex_split": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "rls_complex_merge": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "rls_fista_create_batched": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _pvp]),
    "rls_fista_init_batched": (_i32, [_vp, _vp, _i64, _f, _f, _f, _i32, _i32]),
    "rls_fista_get_status_batched": (_i32, [_vp, C.POINTER(FistaStatus)]),
    "rls_fista_init_local_a": (_i32, [_vp, _vp]),
    "rls_fista_init_local_b": (_i32, [_vp, _f, _f, _f, _i32, _i32]),
    "rls_fista_step_local_a": (_i32, [_vp]),
    "rls_fista_step_local_b": (_i32, [_vp]),
    "rls_cg_local_apply": (_i32, [_vp, _vp]),
    "rls_cg_local_start": (_i32, [_vp, _vp, _vp, _f, _i32, _f]),
    "rls_cg_local_update": (_i32, [_vp, _vp]),
    "rls_comm_create": (_i32, [_i32, C.POINTER(C.c_int32), C.POINTER(_vp), _i32, C.POINTER(_vp)]),
    "rls_comm_destroy": (_i32, [_vp]),
    "rls_comm_size": (_i32, [_vp]),
    "rls_comm_transport": (_i32, [_vp]),
    "rls_comm_peer_access": (_i32, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rls_comm_ctx": (_i32, [_vp, _i32, C.POINTER(_vp)]),
    "rls_comm_sync": (_i32, [_vp]),
    "rls_allreduce_sum": (_i32, [_vp, C.POINTER(_vp), C.c_int64, _i32]),
    "rls_cgnr_init_rowsharded": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.c_float, C.c_float, _i32]),
    "rls_cgnr_step_rowsharded": (_i32, [_vp, C.POINTER(_vp), _i32]),
    "rls_comm_set_threads": (_i32, [_vp, _i32]),
    "rls_comm_debug_busy_seconds": (_i32, [_vp, C.POINTER(C.c_double)]),
    "rls_fista_init_rowsharded": (_i32, [_vp, _pvp, _pvp, _f, _f, _f, _i32, _i32]),
    "rls_fista_step_rowsharded": (_i32, [_vp, _pvp, _i32]),
    "rls_admm_init_rowsharded": (_i32, [_vp, _pvp, _pvp]),
    "rls_admm_step_rowsharded": (_i32, [_vp, _pvp, _i32]),
    "rls_admm_create": (_i32, [_vp, _pvp]),
    "rls_admm_destroy": (_i32, [_vp]),
    "rls_admm_init": (_i32, [_vp, C.POINTER(AdmmParams)]),
    "rls_admm_step": (_i32, [_vp, _i32]),
    "rls_admm_get_status": (_i32, [_vp, C.POINTER(AdmmStatus), _pf, _i32]),
    "rls_admm_step_status": (_i32, [_vp, _i32, C.POINTER(AdmmStatus), _pf, _i32]),
    "rls_admm_get_status_batched": (_i32, [_vp, C.POINTER(AdmmStatus), _pf, _i32]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C regularizedleastsquares.jl_amd/csrc`).  There is no CPU fallback.")
    # torch bundles its own libamdhip64.so under the same soname; whichever HIP runtime is loaded
    # first serves both, so bring torch's in first when torch is installed (two runtimes in one
    # process would not share streams or allocations).
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing only; the library itself links libamdhip64 directly
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, status, what=""):
    if status != 0:
        msg = ""
        if ctx_handle:
            raw = load().rls_last_error_string(ctx_handle)
            msg = raw.decode() if raw else ""
        raise RLSError(f"{what} failed with status {status}: {msg}")
