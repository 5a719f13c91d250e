"""ctypes binding of the C-ABI engine library (include/sparseeventid_amd.h).

The product path has NO CPU fallback: if libsparseeventid_amd.so is missing or no HIP
device is present the ops raise, loudly.  Build with ``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C sparseeventid_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsparseeventid_amd.so")

_lib = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
vp = ctypes.c_void_p

_SIGNATURES = {
    "sei_last_error": (ctypes.c_char_p, []),
    "sei_abi_version": (ctypes.c_int, []),
    "sei_metadata_create": (ctypes.c_int, [ctypes.POINTER(vp)]),
    "sei_metadata_destroy": (None, [vp]),
    "sei_input_build": (ctypes.c_int, [vp, vp, ctypes.c_int64, c_i32p, ctypes.c_int32, ctypes.c_int32, vp, c_i64p]),
    "sei_input_build_like": (ctypes.c_int, [vp, vp, ctypes.c_int64, c_i32p, ctypes.c_int32, ctypes.c_int32, vp, vp, c_i64p]),
    "sei_larcv_to_scn": (ctypes.c_int, [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, vp,
                                        c_i64p]),
    "sei_input_features_fwd": (ctypes.c_int, [vp, vp, ctypes.c_int32, vp, vp]),
    "sei_input_features_bwd": (ctypes.c_int, [vp, vp, ctypes.c_int32, vp, vp]),
    "sei_level_rows": (ctypes.c_int, [vp, c_i32p, c_i64p]),
    "sei_level_coords": (ctypes.c_int, [vp, c_i32p, vp, vp]),
    "sei_batch_size": (ctypes.c_int, [vp, c_i32p]),
    "sei_subm_rulebook": (ctypes.c_int, [vp, c_i32p, c_i32p, vp, ctypes.POINTER(vp)]),
    "sei_conv_rulebook": (ctypes.c_int, [vp, c_i32p, c_i32p, c_i32p, vp, ctypes.POINTER(vp), c_i32p]),
    "sei_rulebook_info": (ctypes.c_int, [vp, c_i64p]),
    "sei_metadata_prebuild_like": (ctypes.c_int, [vp, vp, vp]),
    "sei_metadata_prebuild_forward_like": (ctypes.c_int, [vp, vp, vp]),
    "sei_rulebook_build": (ctypes.c_int, [vp, ctypes.c_int32, vp]),
    "sei_rulebook_export": (ctypes.c_int, [vp, c_i64p, vp, vp, vp]),
    "sei_conv_forward": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp]),
    "sei_conv_backward_input": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp]),
    "sei_conv_backward_input_accumulate": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp]),
    "sei_conv_backward_weight": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp]),
    "sei_conv_backward": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp]),
    "sei_set_backward_streams": (ctypes.c_int, [ctypes.c_int32]),
    "sei_set_product_mode": (ctypes.c_int, [ctypes.c_int32]),
    "sei_backward_join": (ctypes.c_int, [vp]),
    "sei_debug_force_conv_kernel": (ctypes.c_int, [ctypes.c_int32]),
    "sei_debug_force_dw_kernel": (ctypes.c_int, [ctypes.c_int32]),
    "sei_debug_band_timeouts": (ctypes.c_int, [c_i32p, ctypes.c_int32]),
    "sei_bn_forward": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_int32, ctypes.c_float, vp, vp, vp, vp, vp]),
    "sei_conv_bn_forward": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_int32, ctypes.c_float, vp, vp, vp, vp, vp]),
    "sei_conv_bn_inference": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_float,
                                             ctypes.c_float, vp, vp, vp]),
    "sei_bn_backward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int32, vp, vp, vp, ctypes.c_float,
                                       ctypes.c_int32, vp, vp, vp, vp, vp]),
    "sei_bn_backward_colsum": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_float,
                                              ctypes.c_int32, vp, vp, vp, vp, vp, vp]),
    "sei_leaky_forward": (ctypes.c_int, [vp, vp, ctypes.c_int64, ctypes.c_float, vp]),
    "sei_leaky_backward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_float, vp]),
    "sei_add_forward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, vp]),
    "sei_add_leaky_forward": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_float, vp]),
    "sei_sparse_to_dense_forward": (ctypes.c_int, [vp, c_i32p, vp, ctypes.c_int32, vp, vp]),
    "sei_sparse_to_dense_backward": (ctypes.c_int, [vp, c_i32p, vp, ctypes.c_int32, vp, vp]),
    "sei_pool_tanh_forward": (ctypes.c_int, [vp, c_i32p, vp, ctypes.c_int32, vp, vp]),
    "sei_pool_tanh_backward": (ctypes.c_int, [vp, c_i32p, vp, vp, ctypes.c_int32, vp, vp]),
    "sei_prof_enable": (ctypes.c_int, [ctypes.c_int32]),
    "sei_prof_reset": (ctypes.c_int, []),
    "sei_prof_query": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]),
    "sei_debug_sync_count": (ctypes.c_int, [c_i64p, ctypes.c_int32]),
    "sei_pool_stats": (ctypes.c_int, [c_i64p, c_i64p]),
    "sei_pool_trim": (ctypes.c_int, []),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """Load (once) and return the engine library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "sparseeventid_amd: %s is missing -- the HIP engine is not built and there is no CPU "
                "fallback.  Run `make -C sparseeventid_amd/csrc` (needs hipcc)." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().sei_last_error().decode("utf-8", "replace"))


def i32x3(v):
    return (ctypes.c_int32 * 3)(int(v[0]), int(v[1]), int(v[2]))
