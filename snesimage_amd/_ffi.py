"""ctypes binding of libsnesimage_hip.so (the C ABI declared in include/snesimage_hip.h).

The library is built in-tree by `make -C snesimage_amd/csrc` (or __graft_entry__.build()).
There is no fallback: if the shared object is missing, importing the binding raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libsnesimage_hip.so")

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)



class CallResult(C.Structure):
    """snesimage_call_result: what snesimage_last_step reports after one optimizer call."""
    _fields_ = [("error", C.c_double), ("best_k", C.c_int32), ("rgb5", C.c_uint8 * 3), ("changed", C.c_uint8)]


class RunStats(C.Structure):
    """snesimage_run_stats"""
    _fields_ = [("calls", C.c_uint32), ("accepted", C.c_uint32), ("windows", C.c_uint32), ("voided", C.c_uint32),
                ("scored", C.c_uint64), ("useful", C.c_uint64)]


# every symbol include/snesimage_hip.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("snesimage_create", C.c_int32, [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                                     C.POINTER(C.c_void_p)]),
    ("snesimage_destroy", None, [C.c_void_p]),
    ("snesimage_set_stream", C.c_int32, [C.c_void_p, C.c_void_p]),
    ("snesimage_sync", C.c_int32, [C.c_void_p]),
    ("snesimage_set_chunk", C.c_int32, [C.c_void_p, C.c_uint32]),
    ("snesimage_initialize_tiles", C.c_int32, [C.c_void_p]),
    ("snesimage_recalculate_palettes", C.c_int32, [C.c_void_p]),
    ("snesimage_optimize", C.c_int32, [C.c_void_p]),
    ("snesimage_error", C.c_int32, [C.c_void_p, _f64p]),
    ("snesimage_reassign_tiles", C.c_int32, [C.c_void_p, _u32p]),
    ("snesimage_score_candidates", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, _f64p]),
    ("snesimage_score_candidates_device", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                                      C.c_void_p, C.c_void_p]),
    ("snesimage_remap_candidates_device", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("snesimage_step", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.c_uint64, C.c_uint32, _f64p, _u8p]),
    ("snesimage_step_async", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                         C.c_uint64, C.c_uint32]),
    ("snesimage_last_step", C.c_int32, [C.c_void_p, _f64p, _u8p, _i32p]),
    ("snesimage_step_begin", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                         C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("snesimage_step_commit", C.c_int32, [C.c_void_p, C.c_void_p]),
    ("snesimage_run_slots", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, _u32p, _u32p, _u32p, _u32p, C.c_uint32, C.c_uint32,
                                        C.POINTER(CallResult), C.POINTER(RunStats)]),
    ("snesimage_slots_reserve", C.c_int32, [C.c_void_p, C.c_uint32]),
    ("snesimage_slots_begin", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, _u32p, _u32p]),
    ("snesimage_slots_commit", C.c_int32, [C.c_void_p, C.c_void_p, _u32p, _u32p, C.POINTER(CallResult)]),
    ("snesimage_group_create", C.c_int32, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p)]),
    ("snesimage_group_destroy", None, [C.c_void_p]),
    ("snesimage_group_step", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32,
                                         _f64p, _u8p]),
    ("snesimage_group_run_slots", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, _u32p, _u32p, _u32p, _u32p, C.c_uint32, C.c_uint32,
                                              C.POINTER(CallResult), C.POINTER(RunStats)]),
    ("snesimage_batch_create", C.c_int32, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p)]),
    ("snesimage_batch_destroy", None, [C.c_void_p]),
    ("snesimage_batch_step_async", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                                               C.c_uint64, C.c_uint32]),
    ("snesimage_batch_sync", C.c_int32, [C.c_void_p]),
    ("snesimage_get_tile_palettes", C.c_int32, [C.c_void_p, _u8p]),
    ("snesimage_set_tile_palettes", C.c_int32, [C.c_void_p, _u8p]),
    ("snesimage_get_palette_rgb5", C.c_int32, [C.c_void_p, _u8p]),
    ("snesimage_set_palette_rgb5", C.c_int32, [C.c_void_p, _u8p]),
    ("snesimage_get_palette_u16", C.c_int32, [C.c_void_p, _u16p]),
    ("snesimage_get_palette_map", C.c_int32, [C.c_void_p, _u8p]),
    ("snesimage_set_palette_map", C.c_int32, [C.c_void_p, _u8p]),
    ("snesimage_as_rgba", C.c_int32, [C.c_void_p, _u8p]),
    ("snesimage_as_json", C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("snesimage_random_candidates", None, [C.c_uint64, C.c_uint64, C.c_uint32, _u8p]),
    ("snesimage_schedule_next", None, [C.c_uint32, C.c_uint32, C.c_int32, _u32p, _u32p, _u32p, _u32p, _u32p]),
    ("snesimage_debug_math", C.c_int32, [C.c_int32, C.c_int32, _f32p, _f32p, C.c_uint32, _f32p]),
    ("snesimage_debug_fail_alloc", None, [C.c_int32]),
    ("snesimage_timing_enable", C.c_int32, [C.c_void_p, C.c_int32]),
    ("snesimage_timing_read", C.c_int32, [C.c_void_p, _f64p, _u64p, _u64p]),
    ("snesimage_last_error", C.c_char_p, []),
    ("snesimage_version", C.c_char_p, []),
]

_lib = None


def load():
    """Load libsnesimage_hip.so and type every entry point. Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "libsnesimage_hip.so not found at %s: build it with `make -C snesimage_amd/csrc` "
                "(there is no CPU fallback)" % SO_PATH)
        lib = C.CDLL(SO_PATH)
        for name, res, args in SIGNATURES:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
