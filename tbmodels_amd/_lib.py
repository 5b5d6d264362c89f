"""
ctypes binding of ``libtbk.so`` (C ABI: ``include/tbk.h``).

There is deliberately no fallback: if the library is missing or no GPU is visible, the compute
entry points raise.  Importing this module does not load the library -- :func:`lib` does, on first
use -- so the pure-host parts of the package (model construction, synthetic generators, argument
checks) work on a machine without a GPU or a build.
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TBK_LIBTBK: another build of the same library -- tools/ and one test load `libtbk_experiments.so`, the build whose
# measurement switches read the environment (`make -C tbmodels_amd/csrc EXPERIMENTS=1`); the product never sets it)
LIB_PATH = os.environ.get("TBK_LIBTBK") or os.path.join(_HERE, "libtbk.so")
EXPERIMENTS_LIB_PATH = os.path.join(_HERE, "libtbk_experiments.so")

TBK_OK = 0
TBK_ERR_ARGUMENT = 1
TBK_ERR_DEVICE = 2
TBK_ERR_MEMORY = 3
TBK_ERR_NOT_FINITE = 4
TBK_ERR_NO_CONVERGENCE = 5

TBK_EIG_AUTO, TBK_EIG_WAVE, TBK_EIG_ROCSOLVER = 0, 1, 2
TBK_OPT_EIGENSOLVER, TBK_OPT_K_CHUNK, TBK_OPT_TIMING, TBK_OPT_FOLD = 1, 2, 3, 4
TBK_REDUCE_AUTO, TBK_REDUCE_ONE_STAGE, TBK_REDUCE_TWO_STAGE = 0, 1, 2
TBK_CNT_EIGENVAL_CALLS, TBK_CNT_FOLDED_CALLS, TBK_CNT_FOLDED_KPOINTS, TBK_CNT_LIBRARY_CALLS = 0, 1, 2, 3
TBK_T_PHASE, TBK_T_HK, TBK_T_EIG, TBK_T_QL, TBK_T_COUNT = 0, 1, 2, 3, 4
STAGE_NAMES = ("phase", "hk", "eig", "ql")

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)

#: name -> (restype, argtypes); one entry per function declared in include/tbk.h
SIGNATURES = {
    "tbk_version": (ctypes.c_char_p, []),
    "tbk_last_error": (ctypes.c_char_p, []),
    "tbk_device_count": (_c_int, [ctypes.POINTER(_c_int)]),
    "tbk_model_create_dense": (_c_int, [_c_int, _c_int, _c_int, _c_i64, _vp, _vp, _pp]),
    "tbk_model_create_csr": (_c_int, [_c_int, _c_int, _c_int, _c_i64, _vp, _vp, _vp, _vp, _vp, _pp]),
    "tbk_model_destroy": (None, [_vp]),
    "tbk_model_set_option": (_c_int, [_vp, _c_int, _c_i64]),
    "tbk_model_info": (
        _c_int,
        [_vp, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.POINTER(_c_i64),
         ctypes.POINTER(_c_int), ctypes.POINTER(_c_i64)],
    ),
    "tbk_hamilton": (_c_int, [_vp, _vp, _c_i64, _c_int, _vp, _vp]),
    "tbk_eigenval": (_c_int, [_vp, _vp, _c_i64, _vp]),
    "tbk_eigenval_multi": (_c_int, [_vp, _c_int, _vp, _c_i64, _vp]),
    "tbk_hamilton_multi": (_c_int, [_vp, _c_int, _vp, _c_i64, _c_int, _vp, _vp]),
    "tbk_hamilton_device": (_c_int, [_vp, _vp, _c_i64, _c_int, _vp, _vp]),
    "tbk_eigenval_device": (_c_int, [_vp, _vp, _c_i64, _vp]),
    "tbk_eigenval_device_hint": (_c_int, [_vp, _vp, _vp, _c_i64, _vp]),
    "tbk_model_counter": (_c_int, [_vp, _c_int, ctypes.POINTER(_c_i64)]),
    "tbk_eigenval_check": (_c_int, [_vp]),
    "tbk_synchronize": (_c_int, [_vp]),
    "tbk_tridiagonal_reduce": (_c_int, [_c_int, _c_int, _c_i64, _vp, _c_int, _vp, _vp, _vp]),
    "tbk_reduce_standalone": (_c_int, [_c_int, _c_int, _c_i64, _c_int, ctypes.POINTER(ctypes.c_double)]),
    "tbk_kdotp_create": (_c_int, [_c_int, _c_int, _c_int, _c_i64, _vp, _vp, _pp]),
    "tbk_kdotp_destroy": (None, [_vp]),
    "tbk_kdotp_hamilton": (_c_int, [_vp, _vp, _c_i64, _vp]),
    "tbk_kdotp_eigenval": (_c_int, [_vp, _vp, _c_i64, _vp]),
    "tbk_kdotp_eigenval_multi": (_c_int, [_vp, _c_int, _vp, _c_i64, _vp]),
    "tbk_kdotp_hamilton_multi": (_c_int, [_vp, _c_int, _vp, _c_i64, _vp]),
    "tbk_kdotp_coefficients": (_c_int, [_vp, _vp, _c_i64, _vp, _vp, _vp]),
    "tbk_device_malloc": (_c_int, [_c_int, _c_i64, _pp]),
    "tbk_device_free": (_c_int, [_c_int, _vp]),
    "tbk_memcpy_h2d": (_c_int, [_c_int, _vp, _vp, _c_i64]),
    "tbk_memcpy_d2h": (_c_int, [_c_int, _vp, _vp, _c_i64]),
    "tbk_device_mem_info": (_c_int, [_c_int, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)]),
    "tbk_get_timing": (_c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_i64), _c_int]),
    "tbk_comm_unique_id": (_c_int, [_vp]),
    "tbk_comm_create": (_c_int, [_c_int, _c_int, _c_int, _vp, _pp]),
    "tbk_comm_destroy": (None, [_vp]),
    "tbk_comm_ranks": (_c_int, [_vp, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "tbk_comm_allgather_f64": (_c_int, [_vp, _vp, _vp, _vp, _c_i64]),
    "tbk_comm_allgather_f64_overlapped": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_int]),
    "tbk_comm_wait_slot": (_c_int, [_vp, _vp, _c_int]),
    "tbk_comm_synchronize": (_c_int, [_vp]),
    "tbk_comm_agree": (_c_int, [_vp, _c_int, _vp]),
    "tbk_comm_prepare_gather": (_c_int, [_vp, _c_int, _c_i64]),
    "tbk_eigenval_device_gather": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_i64, _c_int, _vp, _vp]),
    "tbk_mfma_f64_peak": (_c_int, [_c_int, ctypes.POINTER(ctypes.c_double)]),
}

_LIB = None


class TbkLibraryError(RuntimeError):
    """``libtbk.so`` is missing or cannot be loaded: the product path has no CPU fallback."""


def lib():
    """The loaded ``libtbk.so`` (loads it on first call; raises :class:`TbkLibraryError` if absent)."""
    global _LIB  # pylint: disable=global-statement
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TbkLibraryError(
                "{} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C tbmodels_amd/csrc`). There is no CPU fallback.".format(LIB_PATH)
            )
        try:
            handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        except OSError as exc:
            raise TbkLibraryError("cannot load {}: {}".format(LIB_PATH, exc)) from exc
        for name, (restype, argtypes) in SIGNATURES.items():
            func = getattr(handle, name)
            func.restype = restype
            func.argtypes = argtypes
        _LIB = handle
    return _LIB


def last_error():
    msg = lib().tbk_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status):
    """Map a ``tbk_status`` to the exception the reference's callers would see for the same condition."""
    if status == TBK_OK:
        return
    msg = last_error()
    if status in (TBK_ERR_ARGUMENT, TBK_ERR_NOT_FINITE):
        raise ValueError(msg)
    if status == TBK_ERR_MEMORY:
        raise MemoryError(msg)
    if status == TBK_ERR_NO_CONVERGENCE:
        raise np.linalg.LinAlgError(msg)  # what scipy.linalg.eigvalsh raises
    raise RuntimeError(msg)


def ptr(array):
    """Address of a C-contiguous numpy array as an ``int`` (or None): what a ``void*`` parameter of the ctypes signatures takes.
    (``array.ctypes.data_as(c_void_p)`` builds two ctypes objects per call: 2.1 us against 0.9 -- three pointers per one-k call.)"""
    if array is None:
        return None
    assert array.flags.c_contiguous
    return array.__array_interface__["data"][0]


def device_count():
    count = ctypes.c_int(0)
    check(lib().tbk_device_count(ctypes.byref(count)))
    return count.value
