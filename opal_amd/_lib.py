"""ctypes binding of libopal_hip.so (include/opal_hip.h).  There is no fallback: if the library is missing or a
call fails, this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libopal_hip.so")

OPAL_OK = 0
OPAL_EINVAL = -22
OPAL_ENOMEM = -12
OPAL_ESTATE = -1
OPAL_EHIP = -5
OPAL_ERANGE = -34
RAW_FLAG = 0x80000000
VW_CONSTANT = 11650396

# every symbol include/opal_hip.h declares: (restype, argtypes)
_vp, _u64, _u32, _i32, _int, _f32, _f64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int, C.c_float, C.c_double
_pp = C.POINTER(C.c_void_p)
PROTOTYPES = {
    "opal_version": (C.c_char_p, []),
    "opal_last_error": (C.c_char_p, [_vp]),
    "opal_ctx_create": (_int, [_int, _vp, _pp]),
    "opal_ctx_destroy": (_int, [_vp]),
    "opal_sync": (_int, [_vp]),
    "opal_ctx_stream": (_vp, [_vp]),
    "opal_set_patterns": (_int, [_vp, _vp, _int, _int, _int]),
    "opal_load_weights": (_int, [_vp, _vp, _u64, _int, _int]),
    "opal_set_weights_device": (_int, [_vp, _vp, _u64, _int, _int]),
    "opal_adopt_weights_device": (_int, [_vp, _vp, _u64, _u64, _int, _int]),
    "opal_set_tuning": (_int, [_vp, _int, _u64, _int, _u64]),
    "opal_set_vw_options": (_int, [_vp, _int, _int, _int, _f32, _f32]),
    "opal_pack_reads": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _vp, C.POINTER(_u64)]),
    "opal_rng_create": (_int, [_pp]),
    "opal_rng_destroy": (_int, [_vp]),
    "opal_rng_seed": (_int, [_vp, _vp, _int]),
    "opal_rng_randbelow": (_int, [_vp, _u64, C.POINTER(_u64)]),
    "opal_sample_positions": (_int, [_vp, _vp, _u64, _u64, _f64, _vp, _vp, _u64, C.POINTER(_u64)]),
    "opal_rng_shuffle": (_int, [_vp, _vp, _u64]),
    "opal_feature_offsets": (_int, [_vp, _vp, _u64, _int, _vp]),
    "opal_feature_offsets_noalloc": (_int, [_vp, _vp, _u64, _int, _vp]),
    "opal_fragment_views": (_int, [_vp, _vp, _vp, _u64, _vp, _vp, _u64, C.c_uint32, _vp, _vp]),
    "opal_features": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _int, _int, _vp, _vp, _u64]),
    "opal_feature_names": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _int, _vp, _vp, _u64]),
    "opal_predict": (_int, [_vp, _vp, _u64, _vp, _vp, _u64, _int, _vp, _vp]),
    "opal_predict_kernel_name": (C.c_char_p, [_vp]),
    "opal_ingest_scan": (_int, [_vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "opal_ingest_fasta": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "opal_eval_counts": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    "opal_train_begin": (_int, [_vp, _int, _int, _int]),
    "opal_train_batch": (_int, [_vp, _vp, _u64, _vp, _vp, _vp, _u64, _int, _f32, _f32, _f32, _int, _vp]),
    "opal_get_weights": (_int, [_vp, _vp, _int]),
    "opal_load_weights_sparse": (_int, [_vp, _vp, _vp, _u64, _int, _int]),
    "opal_get_accumulators": (_int, [_vp, _vp, _int]),
    "opal_load_accumulators_sparse": (_int, [_vp, _vp, _vp, _u64]),
}

_lib = None


class OpalError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libopal_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Loads libopal_hip.so (built by __graft_entry__.build()); raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # torch wheels bundle their own libamdhip64.so.7; loading torch FIRST makes the dynamic loader hand that
        # same runtime to libopal_hip.so (same SONAME).  Two HIP runtimes in one process cannot both see the GPU.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, ctx=None):
    if rc != OPAL_OK:
        msg = lib().opal_last_error(ctx)
        raise OpalError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
