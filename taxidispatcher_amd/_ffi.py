"""ctypes binding of include/taxidispatcher_amd.h (the C-ABI HIP library).

There is NO CPU fallback: if the shared library is missing or no MI355X is visible, calls
raise.  The oracle under oracle/ is test infrastructure and is never imported from here.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TD_LIB") or os.path.join(_HERE, "libtaxidispatcher_amd.so")

c_i32p = ctypes.c_void_p  # raw addresses: host numpy buffers or device pointers
_lib = None
_inited_device = None

TD_K = {"cost_build": 0, "gen": 1, "compress": 2, "bid": 3, "assign": 4, "sap": 5, "final": 6, "lcm": 7, "line": 8, "cert": 9}

# name -> (restype, argtypes) ; must list every TD_API symbol of the header
SIGNATURES = {
    "td_init": (ctypes.c_int, [ctypes.c_int]),
    "td_shutdown": (None, []),
    "td_last_error": (ctypes.c_char_p, []),
    "td_set_stream": (ctypes.c_int, [ctypes.c_void_p]),
    "td_synchronize": (ctypes.c_int, []),
    "td_version": (ctypes.c_int, []),
    "td_cost_build": (ctypes.c_int, [c_i32p, c_i32p, ctypes.c_int, c_i32p, c_i32p, ctypes.c_int, c_i32p, ctypes.c_int,
                                     ctypes.c_int32, ctypes.c_int32, ctypes.c_int, c_i32p]),
    "td_cost_build_rows": (ctypes.c_int, [c_i32p, c_i32p, ctypes.c_int, c_i32p, c_i32p, ctypes.c_int, c_i32p, ctypes.c_int,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p]),
    "td_assign": (ctypes.c_int, [ctypes.c_int, c_i32p, c_i32p, ctypes.POINTER(ctypes.c_int64),
                                 ctypes.POINTER(ctypes.c_int64)]),
    "td_set_line_metric": (ctypes.c_int, [ctypes.c_int]),
    "td_solver_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "td_solver_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "td_solver_assign": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_i32p, c_i32p, ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_int64)]),
    "td_solver_build_assign": (ctypes.c_int, [ctypes.c_void_p, c_i32p, ctypes.c_int, c_i32p, ctypes.c_int, c_i32p, ctypes.c_int, ctypes.c_int32,
                                              ctypes.c_int32, c_i32p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "td_build_assign": (ctypes.c_int, [c_i32p, ctypes.c_int, c_i32p, ctypes.c_int, c_i32p, ctypes.c_int, ctypes.c_int32, ctypes.c_int32,
                                       c_i32p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "td_expand_x": (ctypes.c_int, [ctypes.c_int, c_i32p, ctypes.c_void_p]),
    "td_lcm": (ctypes.c_int, [ctypes.c_int, c_i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int32,
                              ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_i32p, c_i32p,
                              ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64),
                              ctypes.POINTER(ctypes.c_int32)]),
    "td_lcm_shard_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p, ctypes.c_int, ctypes.c_int32,
                                           ctypes.POINTER(ctypes.c_void_p)]),
    "td_lcm_shard_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "td_lcm_shard_local_min": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "td_lcm_shard_take": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "td_lcm_shard_round_colmin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "td_lcm_shard_round_apply": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "td_lcm_shard_round_commit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "td_tick": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                               ctypes.c_int32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                               ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                               ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "td_tick_release_workspace": (None, []),
    "td_pool2": (ctypes.c_int, [ctypes.c_int, c_i32p, c_i32p, c_i32p, ctypes.c_int, c_i32p, c_i32p, c_i32p, c_i32p,
                                ctypes.POINTER(ctypes.c_int32)]),
    "td_pool_n": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_int32),
                                 ctypes.POINTER(ctypes.c_int64)]),
    "td_pool_merge": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p, ctypes.c_int, ctypes.c_int, c_i32p,
                                     ctypes.POINTER(ctypes.c_int32)]),
    "td_count_sum": (ctypes.c_int, [ctypes.c_int, c_i32p, c_i32p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_int32)]),
    "td_gen_uniform": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int,
                                      ctypes.c_int, c_i32p]),
    "td_shard_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p)]),
    "td_shard_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "td_shard_compress": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "td_shard_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "td_shard_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "td_shard_keys_len": (ctypes.c_int, [ctypes.c_void_p]),
    "td_shard_bid": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "td_shard_apply": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "td_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "td_comm_init": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "td_comm_destroy": (ctypes.c_int, []),
    "td_shard_rounds": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "td_shard_cc": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]),
    "td_shard_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    "td_shard_owner": (ctypes.c_int, [ctypes.c_void_p, c_i32p, ctypes.c_int]),
    "td_shard_price": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "td_shard_total": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "td_shard_row_to_col": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "td_shard_const_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "td_shard_options": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "td_shard_compress_spec": (ctypes.c_int, [ctypes.c_void_p]),
    "td_shard_total_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "td_shard_blocks_pending": (ctypes.c_int, [ctypes.c_void_p]),
    "td_shard_phase_a": (ctypes.c_int, [ctypes.c_void_p]),
    "td_shard_state_words": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "td_shard_state_export": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "td_shard_state_import": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_int64)]),
    "td_shard_place_const": (ctypes.c_int, [ctypes.c_void_p]),
    "td_set_blocks": (ctypes.c_int, [ctypes.c_int]),
    "td_line_shard_ws_words": (ctypes.c_int64, [ctypes.c_int]),
    "td_line_shard_phase": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "td_line_shard_result": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]),
    "td_ipc_export": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "td_ipc_open": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "td_ipc_close": (ctypes.c_int, [ctypes.c_void_p]),
    "td_memcpy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "td_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "td_profile_get": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_int64)]),
    "td_profile_reset": (ctypes.c_int, []),
    "td_last_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.c_int]),
}


class TdError(RuntimeError):
    """Raised for every non-zero return code of the C ABI (the reference raises on a failed
    solve too: solver.py:38 indexes x=None)."""


def load():
    """dlopen the HIP library (no GPU needed for this step) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own HIP runtime; load it FIRST so the process holds one runtime
        # (the other order leaves torch.cuda unusable: "No HIP GPUs are available").
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise TdError(
            "HIP extension not built: %s is missing. Run `python -m taxidispatcher_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().td_last_error()
        raise TdError("taxidispatcher_amd error %d: %s" % (rc, msg.decode() if msg else "?"))


def init(device=None):
    """Initialise the library on `device` (default: LOCAL_RANK or 0). Raises without a GPU."""
    global _inited_device
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _inited_device == device:
        return lib
    check(lib.td_init(int(device)))
    _inited_device = device
    return lib


def shutdown():
    global _inited_device
    if _lib is not None and _inited_device is not None:
        _lib.td_shutdown()
    _inited_device = None


def lib():
    return init(_inited_device)


def addr(a):
    """Address of a numpy array (host) / torch tensor (host or device) / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise TdError("array handed to the C ABI must be C-contiguous")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        # the ABI reads plain row-major memory: a strided view (or a tensor of another width where
        # int32 is expected) would be read as garbage, not refused, by the kernels
        if hasattr(a, "is_contiguous") and not a.is_contiguous():
            raise TdError("tensor handed to the C ABI must be contiguous")
        # The library works on its own HIP stream (td_set_stream changes that). A CUDA tensor may
        # still be being written by kernels queued on torch's current stream, and torch may hand
        # its memory to the next tensor the moment it is released: fence torch's stream before
        # the library sees the pointer.
        if getattr(a, "is_cuda", False):
            import torch
            torch.cuda.current_stream(a.device).synchronize()
        return a.data_ptr()
    raise TypeError("cannot take the address of %r" % type(a))


def as_i32(a):
    """Host int32 C-contiguous view/copy of an array-like (lists of Python ints or floats that
    hold integers, as the reference passes)."""
    arr = np.asarray(a)
    if arr.dtype != np.int32:
        if arr.dtype.kind == "f":
            r = np.rint(arr)
            if not np.array_equal(r, arr):
                raise TdError("costs / positions must be integer-valued")
            arr = r
        arr = arr.astype(np.int64)
        if arr.size and (arr.max() > 2**31 - 1 or arr.min() < -2**31):
            raise TdError("value out of int32 range")
        arr = arr.astype(np.int32)
    return np.ascontiguousarray(arr)
