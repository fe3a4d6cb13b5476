"""ctypes binding of libmm2chain_hip.so (include/mm2chain.h).  No fallback: a missing library raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MM2C_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libmm2chain_hip.so")   # MM2C_LIB_PATH: an experimental build (tools/probe_prices.sh)

MM2C_F_IGNORE_SEG = 0x1
MM2C_F_FORCE_GENERAL = 0x2


class Params(C.Structure):
    """mm2c_params_t"""
    _fields_ = [("max_dist_x", C.c_int32), ("max_dist_y", C.c_int32), ("bw", C.c_int32),
                ("max_skip", C.c_int32), ("max_iter", C.c_int32), ("gap_scale", C.c_float),
                ("is_cdna", C.c_int32), ("n_segs", C.c_int32), ("q_span_override", C.c_int32),
                ("flags", C.c_int32)]


class Stream(C.Structure):
    """mm2c_stream_t"""
    _fields_ = [("par", Params), ("min_cnt", C.c_int32), ("min_sc", C.c_int32), ("n_tasks", C.c_int64), ("total", C.c_int64),
                ("offsets", C.POINTER(C.c_int64)), ("anchors", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("tasks", C.c_uint64), ("anchors", C.c_uint64), ("launches", C.c_uint64), ("segments", C.c_uint64), ("host_call_ns", C.c_uint64), ("passes", C.c_uint64)]


class SlotStats(C.Structure):
    """mm2c_slot_stats_t"""
    _fields_ = [("device", C.c_int32), ("reserved", C.c_int32), ("passes", C.c_uint64), ("calls", C.c_uint64), ("anchors", C.c_uint64), ("declined", C.c_uint64)]


class StageStats(C.Structure):
    """mm2c_stage_stats_t"""
    _fields_ = [(k, C.c_uint64) for k in ("calls", "chunks", "total_ns", "alloc_ns", "n_alloc", "free_ns", "n_free", "setup_ns", "h2d_ns", "seed_ns", "dp_ns",
                                          "epi_ns", "d2h_ns", "wait_ns")]


# every symbol include/mm2chain.h declares with C linkage: name -> (restype, argtypes)
C_SYMBOLS = {
    "mm2c_init": (C.c_int, [C.c_int]),
    "mm2c_init_devices": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "mm2c_init_async": (C.c_int, [C.c_int]),
    "mm2c_init_wait": (C.c_int, []),
    "mm2c_warm_up": (C.c_int, []),
    "mm2c_device_count": (C.c_int, []),
    "mm2c_numa_cpulist": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "mm2c_slot_worker_node": (C.c_int, [C.c_int]),
    "mm2c_split_tasks": (C.c_int, [C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "mm2c_shutdown": (None, []),
    "mm2c_last_error": (C.c_char_p, []),
    "mm2c_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "mm2c_device_identity": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "mm2c_tune": (C.c_int, [C.c_char_p, C.c_int]),
    "mm2c_split_model": (C.c_int, [C.c_char_p] + [C.POINTER(C.c_float)] * 5),
    "mm2c_params_map_ont": (None, [C.POINTER(Params)]),
    "mm2c_params_fpga_v2": (None, [C.POINTER(Params), C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mm2c_plan_create": (C.c_void_p, [C.POINTER(Params), C.c_int64, C.c_void_p]),
    "mm2c_plan_destroy": (None, [C.c_void_p]),
    "mm2c_plan_total_anchors": (C.c_int64, [C.c_void_p]),
    "mm2c_plan_run_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_plan_set_device_offsets": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mm2c_plan_run_device_n": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "mm2c_plan_predict_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_plan_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mm2c_plan_last_variant": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "mm2c_plan_last_route": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mm2c_route_pieces": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "mm2c_last_host_variant": (C.c_int, [C.c_char_p, C.c_size_t]),
    "mm2c_debug_label_hits": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "mm2c_plan_last_prepass_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mm2c_chain_batch_host": (C.c_int, [C.POINTER(Params), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_pinned_alloc": (C.c_void_p, [C.c_size_t]),
    "mm2c_pinned_free": (None, [C.c_void_p]),
    "mm2c_chain_task_host": (C.c_int, [C.POINTER(Params), C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int]),
    "mm2c_chain_task_host_pred": (C.c_int, [C.POINTER(Params), C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]),
    "mm2c_get_slot_stats": (C.c_int, [C.c_int, C.POINTER(SlotStats)]),
    "mm2c_route_slot": (C.c_int, [C.c_int, C.c_void_p, C.c_int]),
    "mm_chain_dp": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                 C.c_int64, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_void_p, C.c_int]),
    "mm2c_plan_chains_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_plan_chains_device_n": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "mm2c_plan_last_epilogue_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mm2c_chain_epilogue_host": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_mm_chain_dp_batch_host": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_seedplan_create": (C.c_void_p, [C.c_int64, C.c_void_p, C.c_void_p]),
    "mm2c_seedplan_destroy": (None, [C.c_void_p]),
    "mm2c_seedplan_run_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_seedplan_run_device_n": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                             C.c_void_p]),
    "mm2c_seedplan_run_device_skip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                C.c_int64, C.c_void_p, C.c_void_p]),
    "mm2c_seedplan_set_heap_sort": (C.c_int, [C.c_void_p, C.c_int]),
    "mm2c_seedplan_check": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mm2c_seedplan_last_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "mm2c_seed_hits_batch_host": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_seed_chain_batch_host": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_seed_chain_batch_pool": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm2c_hitpool_create": (C.c_void_p, [C.c_void_p, C.c_int64]),
    "mm2c_hitpool_size": (C.c_int64, [C.c_void_p]),
    "mm2c_hitpool_destroy": (None, [C.c_void_p]),
    "mm2c_get_stats": (None, [C.POINTER(Stats)]),
    "mm2c_get_stage_stats": (None, [C.POINTER(StageStats)]),
    "mm2c_reset_stage_stats": (None, []),
    "mm2c_stream_write": (C.c_int, [C.c_char_p, C.POINTER(Params), C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "mm2c_stream_read": (C.c_int, [C.c_char_p, C.POINTER(Stream)]),
    "mm2c_stream_from_seed_dump": (C.c_int, [C.c_char_p, C.POINTER(Params), C.c_int, C.c_int, C.POINTER(Stream)]),
    "mm2c_stream_free": (None, [C.POINTER(Stream)]),
}
# C++-linkage drop-in symbols the reference objects import (chain_hardware.h:68-71)
CXX_SYMBOLS = {
    "_Z18run_chaining_on_hwliiiifP7mm128_tPiS1_Phliff": (C.c_int, [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                                                   C.c_int, C.c_float, C.c_float]),
    "_Z13hardware_initlPc": (C.c_bool, [C.c_long, C.c_char_p]),
    "_Z7cleanupv": (None, []),
    "_Z10checkErroriNSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEE": (None, [C.c_int, C.c_void_p]),   # chain_hardware.h:72 (a std::string by value: not callable from ctypes, only checked for presence)
}

_lib = None


def load():
    """dlopen the in-tree library and type every entry point; raises if it is absent (no CPU path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `make -C minimap2-fpga_amd` (or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in {**C_SYMBOLS, **CXX_SYMBOLS}.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class SeedSkip(C.Structure):
    """mm2c_seed_skip_t"""
    _fields_ = [("flag", C.c_int32), ("d_ref_rank", C.c_void_p), ("d_ref_len", C.c_void_p), ("d_q_lo", C.c_void_p), ("d_q_eq", C.c_void_p)]


class Mm2cError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().mm2c_last_error()
        raise Mm2cError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
