"""ctypes binding of libomok_hip.so (the C ABI declared in include/omok_hip.h).

There is no CPU fallback: if the shared library is missing and cannot be built, or exports
fewer symbols than the header declares, importing the engine raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libomok_hip.so")
if os.environ.get("AO_LIB_TAG"):   # developer switch: an experiment build made with AO_BUILD_TAG (timing knock-outs)
    LIB_PATH = os.path.join(HERE, "libomok_hip_%s.so" % os.environ["AO_LIB_TAG"])

AO_ROOT_FRESH, AO_ROOT_UNEXPANDED, AO_ROOT_EXPANDED = 0, 1, 2


class AoConfig(C.Structure):
    _fields_ = [("board", C.c_int32), ("win_mark", C.c_int32), ("sims", C.c_int32),
                ("inplanes", C.c_int32), ("games", C.c_int32), ("noise", C.c_int32),
                ("node_cap", C.c_int32), ("device", C.c_int32), ("c_puct", C.c_double),
                ("alpha", C.c_double), ("arena_fraction", C.c_double)]


class AoRolloutConfig(C.Structure):
    _fields_ = [("board", C.c_int32), ("win_mark", C.c_int32), ("sims", C.c_int32), ("games", C.c_int32),
                ("mode", C.c_int32), ("device", C.c_int32), ("c_puct", C.c_double)]


class AoTttConfig(C.Structure):
    _fields_ = [("board", C.c_int32), ("win_mark", C.c_int32), ("sims", C.c_int32), ("games", C.c_int32),
                ("device", C.c_int32)]


_P = C.POINTER
_vp = C.c_void_p
_i32p, _u32p, _u8p, _i8p, _f64p, _i64p = (_P(C.c_int32), _P(C.c_uint32), _P(C.c_uint8),
                                          _P(C.c_int8), _P(C.c_double), _P(C.c_int64))

# name -> (restype, argtypes): every entry point of include/omok_hip.h
SYMBOLS = {
    "ao_version": (C.c_char_p, []),
    "ao_abi_version": (C.c_int, []),
    "ao_create": (C.c_int, [_P(AoConfig), _P(_vp)]),
    "ao_destroy": (None, [_vp]),
    "ao_last_error": (C.c_char_p, [_vp]),
    "ao_stream": (_vp, [_vp]),
    "ao_set_stream": (C.c_int, [_vp, _vp]),
    "ao_sync": (C.c_int, [_vp]),
    "ao_seed": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "ao_seed_all": (C.c_int, [_vp, _u32p]),
    "ao_seed_games": (C.c_int, [_vp, _i32p, _u32p, C.c_int32]),
    "ao_get_rng_state": (C.c_int, [_vp, C.c_int, _u32p, _i32p, _i32p, _f64p]),
    "ao_set_rng_state": (C.c_int, [_vp, C.c_int, _u32p, C.c_int32, C.c_int32, C.c_double]),
    "ao_reset": (C.c_int, [_vp, _u8p]),
    "ao_set_root": (C.c_int, [_vp, C.c_int, _i32p, C.c_int32, _i32p]),
    "ao_set_roots": (C.c_int, [_vp, _u8p, _i32p, C.c_int32, _i32p, _i32p]),
    "ao_begin_move": (C.c_int, [_vp, _u8p]),
    "ao_sims_left": (C.c_int, [_vp]),
    "ao_collect_leaves": (C.c_int, [_vp, _vp]),
    "ao_apply_evals": (C.c_int, [_vp, _vp, _vp]),
    "ao_end_move": (C.c_int, [_vp, _i8p, _f64p, _f64p, _f64p]),
    "ao_play": (C.c_int, [_vp, _i32p, _i32p]),
    "ao_search": (C.c_int, [_vp, _vp, _u8p, _i8p, _f64p, _f64p, _f64p]),
    "ao_get_moves": (C.c_int, [_vp, C.c_int, _i32p, _i32p]),
    "ao_get_root_children": (C.c_int, [_vp, C.c_int, _i32p, _f64p, _f64p, _f64p, _f64p, _i32p]),
    "ao_tree_nodes": (C.c_int, [_vp, C.c_int, _i64p, _i64p]),
    "ao_tree_timing": (C.c_int, [_vp, C.c_int, _f64p, _i64p]),
    "ao_trim_stats": (C.c_int, [_vp, _i64p, _i64p]),
    "ao_node_cap": (C.c_int, [_vp, _i32p, _i32p]),
    "ao_host_threads": (C.c_int, []),
    "ao_fp16_range_events": (C.c_int, [_vp, _i64p, _i64p]),
    "ao_search_stats": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "ao_set_row_cap": (C.c_int, [_vp, C.c_int32]),
    "ao_row_stats": (C.c_int, [_vp, _i64p, _i64p, _i64p, _i64p]),
    "ao_set_eval_log": (C.c_int, [_vp, _i32p, C.c_int32, _vp, C.c_int64]),
    "ao_eval_log_count": (C.c_int, [_vp]),
    "ao_net_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P(_vp)]),
    "ao_net_destroy": (None, [_vp]),
    "ao_net_last_error": (C.c_char_p, [_vp]),
    "ao_net_set_param": (C.c_int, [_vp, C.c_char_p, _P(C.c_float), C.c_int64]),
    "ao_net_finalize": (C.c_int, [_vp]),
    "ao_net_forward": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "ao_net_set_mode": (C.c_int, [_vp, C.c_int]),
    "ao_net_get_mode": (C.c_int, [_vp]),
    "ao_net_status": (C.c_int, [_vp, _vp, _i32p, C.c_int]),
    "ao_net_conv_timing": (C.c_int, [_vp, C.c_int, _f64p, _i64p]),
    "ao_net_products": (C.c_int, [_vp, C.c_int32, _i32p, _i32p]),
    "ao_net_dominant_kernel": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, _f64p]),
    "ao_net_plan_kernel": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, _f64p]),
    "ao_replay_create": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, _P(_vp)]),
    "ao_replay_destroy": (None, [_vp]),
    "ao_replay_last_error": (C.c_char_p, [_vp]),
    "ao_replay_size": (C.c_int64, [_vp]),
    "ao_replay_capacity": (C.c_int64, [_vp]),
    "ao_replay_clear": (C.c_int, [_vp]),
    "ao_replay_extend": (C.c_int, [_vp, _P(C.c_float), _f64p, _P(C.c_float), C.c_int64, C.c_int, _vp]),
    "ao_replay_extend_skip": (C.c_int, [_vp, _P(C.c_float), _f64p, _P(C.c_float), C.c_int64, C.c_int, C.c_int64, _vp]),
    "ao_replay_extend_moves": (C.c_int, [_vp, _P(C.c_int16), C.c_int64, C.c_int64, _P(C.c_int32), _P(C.c_int32), _f64p, _P(C.c_float),
                                         C.c_int64, C.c_int, C.c_int64, _vp]),
    "ao_replay_gather": (C.c_int, [_vp, _i64p, C.c_int64, _vp, _vp, _vp, _vp]),
    "ao_replay_read": (C.c_int, [_vp, C.c_int64, C.c_int64, _f64p, _f64p, _f64p]),
    "ao_rollout_create": (C.c_int, [_P(AoRolloutConfig), _P(_vp)]),
    "ao_rollout_destroy": (None, [_vp]),
    "ao_rollout_last_error": (C.c_char_p, [_vp]),
    "ao_rollout_seed": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "ao_rollout_get_rng_state": (C.c_int, [_vp, C.c_int, _u32p, _i32p, _i32p, _f64p]),
    "ao_rollout_set_rng_state": (C.c_int, [_vp, C.c_int, _u32p, C.c_int32, C.c_int32, C.c_double]),
    "ao_rollout_search": (C.c_int, [_vp, _i32p, _i32p, _u8p, _f64p, _f64p, _i32p]),
    "ao_ttt_create": (C.c_int, [_P(AoTttConfig), _P(_vp)]),
    "ao_ttt_destroy": (None, [_vp]),
    "ao_ttt_last_error": (C.c_char_p, [_vp]),
    "ao_ttt_seed": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "ao_ttt_get_rng_state": (C.c_int, [_vp, C.c_int, _u32p, _i32p]),
    "ao_ttt_set_rng_state": (C.c_int, [_vp, C.c_int, _u32p, C.c_int32]),
    "ao_ttt_search": (C.c_int, [_vp, _i8p, _i32p, _u8p, _f64p, _f64p, _i32p]),
}

_lib = None


def load(build_if_missing=True):
    """Returns the loaded library. Raises RuntimeError when it is not available."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64.so (same SONAME); importing it first makes the engine
        # and torch share one HIP runtime, so device pointers and streams can be exchanged.
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise RuntimeError("libomok_hip.so is missing: run `python -m alpha_omok_amd.build`")
        from . import build as _build
        _build.build()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RuntimeError("libomok_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    if lib.ao_abi_version() != 2:
        raise RuntimeError("libomok_hip.so ABI version mismatch")
    _lib = lib
    return lib
