"""ctypes loader of liblocus_hip.so (the C ABI declared in include/locus_hip.h).

There is deliberately NO fallback: if the HIP library is missing or no GPU is
visible, every compute entry point raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from this package.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOCUS_HIP_LIB", os.path.join(_HERE, "liblocus_hip.so"))

c_double_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_dbl = ctypes.c_double
_ptr = ctypes.c_void_p
_int = ctypes.c_int


class LocusHipError(RuntimeError):
    """Non-zero status from liblocus_hip; `.code` holds it."""

    def __init__(self, code, msg):
        super().__init__(f"liblocus_hip error {code}: {msg}")
        self.code = code


class SweepArgs(ctypes.Structure):
    """struct locus_hip_sweep_args (include/locus_hip.h)."""
    _fields_ = [
        ("n", _i64), ("p", _i64), ("d", _i64),
        ("Xp", _ptr), ("xtx", _ptr), ("shuffled_ind", _ptr), ("lo", _ptr), ("l1", _ptr),
        ("scale", _ptr), ("hinv", _ptr), ("cst", _ptr), ("s2k", _ptr), ("cfac", _dbl),
        ("s2", _ptr), ("xwx", _ptr), ("W2", _ptr), ("half_lsi", _dbl),
        ("R", _ptr), ("state", _ptr), ("stats", _ptr),
        ("workspace", _ptr), ("workspace_bytes", _i64), ("nsteps", _i64),
        ("cfg", ctypes.c_int32 * 4),
        ("ls2", _ptr),
        ("prepacked", ctypes.c_int32), ("pad_", ctypes.c_int32),
    ]


class TailArgs(ctypes.Structure):
    """struct locus_hip_tail_args (include/locus_hip.h)."""
    _fields_ = [
        ("n", _i64), ("p", _i64), ("d_loc", _i64), ("d_total", _i64),
        ("a", _ptr), ("b", _ptr), ("eta", _ptr), ("kappa", _ptr), ("lam", _dbl), ("nu", _dbl),
        ("stats", _ptr), ("tau", _ptr), ("s2", _ptr), ("rs_gam", _ptr), ("ent", _ptr),
        ("si_in", _ptr), ("shard2", _ptr), ("scal", _ptr),
        ("with_elbo", ctypes.c_int32), ("pad_", ctypes.c_int32), ("c_next", _dbl),
        ("lo", _ptr), ("l1", _ptr), ("tau_next", _ptr), ("s2_next", _ptr), ("scale", _ptr),
        ("hinv", _ptr), ("cst", _ptr), ("part", _ptr),
    ]


# name -> (restype, argtypes); every symbol include/locus_hip.h declares
_LOOP_COMMON = [_ptr] * 5 + [_dbl]
SIGNATURES = {
    "locus_hip_last_error": (ctypes.c_char_p, []),
    "locus_hip_version": (_int, []),
    "locus_hip_device_count": (_int, []),
    "locus_hip_core_loop": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr, _ptr, _ptr,
                                   _ptr, _ptr, _dbl, _i64, _i64, _i64]),
    "locus_hip_core_z_loop": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr, _ptr,
                                     _ptr, _ptr, _ptr, _ptr, _dbl, _i64, _i64, _i64]),
    "locus_hip_core_logit_loop": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr, _ptr,
                                         _ptr, _ptr, _ptr, _i64, _i64, _i64]),
    "locus_hip_core_probit_loop": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr,
                                          _ptr, _dbl, _ptr, _i64, _i64, _i64]),
    "locus_hip_core_struct_loop": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr,
                                          _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64]),
    "locus_hip_core_batch": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr, _ptr,
                                    _ptr, _ptr, _i64, _i64, _i64]),
    "locus_hip_core_z_batch": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr, _ptr,
                                      _ptr, _ptr, _ptr, _i64, _i64, _i64]),
    "locus_hip_core_probit_batch": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr,
                                           _ptr, _dbl, _i64, _i64, _i64]),
    "locus_hip_packed_ld": (_i64, [_i64]),
    "locus_hip_pack_x_dev": (_int, [_ptr, _i64, _i64, _ptr, _ptr, _ptr]),
    "locus_hip_sweep_workspace_bytes": (_i64, [_i64, _i64]),
    "locus_hip_sweep_dev": (_int, [ctypes.POINTER(SweepArgs), _int, _ptr]),
    "locus_hip_sweep_prepack_dev": (_int, [ctypes.POINTER(SweepArgs), _ptr]),
    "locus_hip_sweep_plan": (_int, [_i64, _i64, ctypes.POINTER(ctypes.c_int32), _int]),
    "locus_hip_sweep_prepack_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "locus_hip_sweep_status_dev": (_int, [_ptr, _i64, _ptr]),
    "locus_hip_sweep_static_dev": (_int, [ctypes.POINTER(SweepArgs), _int, _ptr]),
    "locus_hip_sweep_config": (_int, [_i64, _i64, _int, ctypes.POINTER(ctypes.c_int32)]),
    "locus_hip_residual_dev": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _int, _ptr]),
    "locus_hip_transpose_dev": (_int, [_ptr, _ptr, _i64, _i64, _ptr]),
    "locus_hip_gamma_rows_dev": (_int, [_ptr, _ptr, _ptr, _i64, _i64, _dbl, _ptr]),
    "locus_hip_state_cols_dev": (_int, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr]),
    "locus_hip_gemm_dev": (_int, [_i64, _i64, _i64, _dbl, _ptr, _i64, _i64, _int, _ptr, _i64, _i64,
                                  _int, _dbl, _ptr, _i64, _i64, _ptr]),
    "locus_hip_batch_update_dev": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _dbl, _dbl, _dbl,
                                          _int, _i64, _i64, _ptr]),
    "locus_hip_m2_minus_b2_dev": (_int, [_ptr, _ptr, _dbl, _ptr, _i64, _i64, _ptr]),
    "locus_hip_s2_from_xwx_dev": (_int, [_ptr, _dbl, _ptr, _i64, _ptr]),
    "locus_hip_half_log_dev": (_int, [_ptr, _ptr, _i64, _ptr]),
    "locus_hip_logit_chi_psi_dev": (_int, [_ptr] * 11 + [_i64, _i64, _i64, _ptr]),
    "locus_hip_logit_sig2_alpha_dev": (_int, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr]),
    "locus_hip_alpha_step_dev": (_int, [_ptr] * 7 + [_i64, _i64, _ptr]),
    "locus_hip_probit_w_dev": (_int, [_ptr] * 7 + [_i64, _i64, _ptr]),
    "locus_hip_identity_tail_dev": (_int, [ctypes.POINTER(TailArgs), _int, _ptr]),
    "locus_hip_bin_rows_dev": (_int, [_ptr, _ptr, _dbl, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _dbl,
                                      _ptr]),
    # resident context (host pointers)
    "locus_hip_ctx_create": (_int, [_ptr, _ptr, _i64, _i64, _i64, ctypes.POINTER(_ptr)]),
    "locus_hip_ctx_set_state": (_int, [_ptr, _ptr, _ptr, _ptr]),
    "locus_hip_ctx_set_response": (_int, [_ptr, _ptr, _ptr]),
    "locus_hip_ctx_sweep": (_int, [_ptr, _ptr, _ptr, _dbl, _ptr, _ptr, _ptr, _ptr, _dbl, _dbl,
                                   _ptr, _ptr, ctypes.POINTER(_dbl)]),
    "locus_hip_ctx_stats": (_int, [_ptr, _ptr, _dbl, _ptr, _ptr, ctypes.POINTER(_dbl)]),
    "locus_hip_ctx_get_state": (_int, [_ptr, _ptr, _ptr, _ptr, _ptr]),
    "locus_hip_ctx_destroy": (_int, [_ptr]),
}

_lib = None


def lib():
    """The loaded library; raises if liblocus_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C locus_amd/csrc` (there is no CPU fallback)")
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as the
        # system one).  It must be the copy the process loads first, otherwise torch sees no
        # GPU and device pointers would cross two runtimes: import torch before the dlopen.
        import torch  # noqa: F401
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(code):
    """Raise LocusHipError for a non-zero status."""
    if code != 0:
        msg = lib().locus_hip_last_error()
        raise LocusHipError(code, msg.decode() if msg else "")


def device_count():
    return lib().locus_hip_device_count()


def sweep_config(n, d, logit=False):
    """Kernel geometry (NPL, C, S, G) the library picks for n rows and d resident responses."""
    cfg = (ctypes.c_int32 * 4)()
    check(lib().locus_hip_sweep_config(int(n), int(d), 1 if logit else 0, cfg))
    return tuple(int(c) for c in cfg)
