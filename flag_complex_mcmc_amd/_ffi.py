"""ctypes loader for libfcm.so (C ABI: include/fcm.h).

The library is the product: HIP kernels for gfx950 plus their host glue.  There
is no Python or CPU fallback; if the shared object is missing this module
raises at import of the symbol table, and every compute call fails with
FCM_ERR_NO_DEVICE when no GPU is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FCM_LIB_PATH: a diagnostic build of the same library (tools/run_stamps.sh); never a different implementation
LIB_PATH = os.environ.get("FCM_LIB_PATH") or os.path.join(_HERE, "libfcm.so")

MAX_COUNTS = 16
NSTATS = 18
STAT_NAMES = ("sampled", "accepted", "n_empty", "n_flip", "n_dmove", "sum_k", "count_len", "status",
              "n_cperm", "n_cswap", "n_changes", "n_redo", "n_wide", "n_big", "n_recheck", "n_held", "n_pairs", "n_shared_rows")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_IO, ERR_PANIC, ERR_NOMEM, ERR_INTERNAL = range(9)

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class FcmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfcm error %d: %s" % (code, msg))
        self.code = code


class CBounds(C.Structure):
    _fields_ = [("flag_count_min", C.c_uint64 * (MAX_COUNTS + 1)), ("min_len", C.c_int32),
                ("flag_count_max", C.c_uint64 * (MAX_COUNTS + 1)), ("max_len", C.c_int32)]


class CSamplerConfig(C.Structure):
    _fields_ = [("n_chains", C.c_uint32), ("first_chain_id", C.c_uint32), ("seed", C.c_uint64),
                ("move_weights", C.c_double * 4), ("sample_distance", C.c_uint64),
                ("dim_cap", C.c_int32), ("device", C.c_int32)]


class CSamplerInfo(C.Structure):
    _fields_ = [("n", C.c_uint32), ("row_words", C.c_uint32), ("n_undirected", C.c_uint64),
                ("n_double", C.c_uint64), ("k_max", C.c_uint32), ("k_mean", C.c_double),
                ("bytes_per_chain", C.c_uint64), ("bytes_static", C.c_uint64),
                ("ncounts", C.c_int32), ("lossless", C.c_int32), ("n_chains", C.c_uint32), ("waves_per_chain", C.c_uint32),
                ("sparse_state", C.c_uint32), ("cooperative_clique_kernel", C.c_uint32)]


class CStateInfo(C.Structure):
    _fields_ = [("sample_number", C.c_uint64), ("total_chains", C.c_uint64), ("set_id", C.c_uint64), ("seed", C.c_uint64),
                ("n", C.c_uint32), ("n_chains", C.c_uint32), ("first_chain_id", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


# name -> (restype, argtypes); every symbol include/fcm.h declares
SIGNATURES = {
    "fcm_last_error": (C.c_char_p, []),
    "fcm_version": (C.c_char_p, []),
    "fcm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "flagser_count_unweighted": (C.c_void_p, [C.c_size_t, C.c_size_t, u32p, C.POINTER(C.c_size_t)]),
    "fcm_graph_new_disconnected": (C.c_int, [C.c_uint32, C.POINTER(vp)]),
    "fcm_graph_from_edges": (C.c_int, [C.c_uint32, C.c_uint64, u32p, C.POINTER(vp)]),
    "fcm_graph_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "fcm_graph_destroy": (None, [vp]),
    "fcm_graph_nnodes": (C.c_uint32, [vp]),
    "fcm_graph_nedges": (C.c_uint64, [vp]),
    "fcm_graph_has_edge": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "fcm_graph_set_edge": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int]),
    "fcm_graph_add_edge": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "fcm_graph_remove_edge": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "fcm_graph_edges": (C.c_int, [vp, u32p, C.c_uint64, u64p]),
    "fcm_graph_undirected_edges": (C.c_int, [vp, u32p, C.c_uint64, u64p]),
    "fcm_graph_flagser_count": (C.c_int, [vp, C.c_int, u64p, C.c_int, C.POINTER(C.c_int)]),
    "fcm_read_flag_file": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "fcm_save_flag_file": (C.c_int, [C.c_char_p, vp]),
    "fcm_target_bounds": (C.c_int, [u64p, C.c_int, C.c_double, C.POINTER(CBounds)]),
    "fcm_bounds_calculate": (C.c_int, [vp, u64p, C.c_int, C.POINTER(CBounds), C.c_int, C.POINTER(CBounds), u64p, C.POINTER(C.c_int)]),
    "fcm_bounds_check": (C.c_int, [C.POINTER(CBounds), u64p, C.c_int]),
    "fcm_default_sample_distance": (C.c_uint64, [C.c_uint64]),
    "fcm_sampler_create": (C.c_int, [vp, C.POINTER(CBounds), C.POINTER(CSamplerConfig), C.POINTER(vp)]),
    "fcm_sampler_destroy": (None, [vp]),
    "fcm_sampler_set_stream": (C.c_int, [vp, vp]),
    "fcm_sampler_step": (C.c_int, [vp, C.c_uint64]),
    "fcm_sampler_next": (C.c_int, [vp]),
    "fcm_sampler_sync": (C.c_int, [vp]),
    "fcm_sampler_last_step_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "fcm_sampler_ncounts": (C.c_int, [vp]),
    "fcm_sampler_sample_distance": (C.c_uint64, [vp]),
    "fcm_sampler_get_counts": (C.c_int, [vp, u64p, i32p]),
    "fcm_sampler_get_stats": (C.c_int, [vp, u64p]),
    "fcm_sampler_get_edges": (C.c_int, [vp, C.c_uint32, u32p, C.c_uint64, u64p]),
    "fcm_sampler_get_edgebits": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint64, u64p]),
    "fcm_sampler_get_double_slots": (C.c_int, [vp, C.c_uint32, u32p, C.c_uint64, u64p]),
    "fcm_sampler_get_info": (C.c_int, [vp, C.POINTER(CSamplerInfo)]),
    "fcm_sampler_debug_stamps": (C.c_int, [vp, u64p]),
    "fcm_sampler_get_bounds": (C.c_int, [vp, C.POINTER(CBounds)]),
    "fcm_sampler_edgeset_neighborhood": (C.c_int, [vp, u32p, C.c_uint32, u32p, C.c_uint64, u64p]),
    "fcm_sampler_apply_transition": (C.c_int, [vp, C.c_uint32, u32p, i32p, C.c_uint32, u64p, i32p, u64p, i32p]),
    "fcm_sampler_revert_transition": (C.c_int, [vp, C.c_uint32, u32p, i32p, C.c_uint32, u64p, C.c_int32, u64p, C.c_int32]),
    "fcm_sampler_single_edge_flip": (C.c_int, [vp, C.c_uint32, C.c_uint64, u32p, i32p, C.POINTER(C.c_uint32)]),
    "fcm_sampler_apply_transitions": (C.c_int, [vp, u32p, i32p, u32p, C.c_uint32, u64p, i32p, u64p, i32p, i32p]),
    "fcm_sampler_revert_transitions": (C.c_int, [vp, u32p, i32p, u32p, C.c_uint32, u64p, i32p, u64p, i32p, i32p]),
    "fcm_sampler_single_edge_flips": (C.c_int, [vp, u64p, u32p, i32p, u32p]),
    "fcm_sampler_save_state": (C.c_int, [vp, C.c_char_p, C.c_uint64]),
    "fcm_sampler_load_state": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp), u64p]),
    "fcm_sampler_save_state_shard": (C.c_int, [vp, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64]),
    "fcm_state_file_info": (C.c_int, [C.c_char_p, C.POINTER(CStateInfo)]),
}

_lib = None


def lib():
    """Load libfcm.so.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `make -C flag_complex_mcmc_amd/csrc` "
                "(or __graft_entry__.build()).  There is no fallback path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        L._free = C.CDLL(None).free
        L._free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise FcmError(rc, lib().fcm_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib().fcm_device_count(C.byref(n)))
    return n.value
