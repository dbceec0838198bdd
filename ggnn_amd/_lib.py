"""ctypes binding of libggnn_amd.so (include/ggnn_c.h).

The HIP library is the product: there is no CPU fallback.  Importing this module never builds
anything; a missing library is a hard error with the build command in the message.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libggnn_amd.so")
# GGNN_AMD_LIB names another build of the same library (A/B measurements of kernel variants).  Like
# every other switch it is honoured only while GGNN_TEST_HOOKS=1 is set: a production process that
# merely inherits the variable always loads the in-tree library.
if os.environ.get("GGNN_TEST_HOOKS") == "1" and os.environ.get("GGNN_AMD_LIB"):
    LIB_PATH = os.environ["GGNN_AMD_LIB"]

OK, INVALID_ARGUMENT, INVALID_STATE, OUT_OF_RANGE, OUT_OF_MEMORY, DEVICE_ERROR, UNSUPPORTED, \
    IO_ERROR = range(8)
F32, U8 = 0, 1
CPU, GPU = 0, 1


class GraphConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("N", "D", "KBuild", "KF", "G", "S", "S0", "S0_off", "SG", "SG_off", "N_all",
                 "ST_all")] + [("Bs", C.c_uint32 * 4), ("Ns", C.c_uint32 * 4),
                               ("Ns_offsets", C.c_uint32 * 4), ("STs_offsets", C.c_uint32 * 4)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else int(v)
        return d


class GraphView(C.Structure):
    _fields_ = [("config", GraphConfig), ("graph", C.c_void_p), ("translation", C.c_void_p),
                ("selection", C.c_void_p), ("nn1_stats", C.c_void_p), ("gpu_id", C.c_int)]


class GGNNError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def raise_for_status(status, message):
    """Map C status codes to the exceptions the reference's binding raises
    (nanobind translates std::out_of_range -> IndexError, std::runtime_error -> RuntimeError)."""
    if status == OK:
        return
    if status == OUT_OF_RANGE:
        raise IndexError(message)
    if status == OUT_OF_MEMORY:
        raise MemoryError(message)
    raise GGNNError(status, message)


_u32, _u64, _f32, _int, _vp, _sz = C.c_uint32, C.c_uint64, C.c_float, C.c_int, C.c_void_p, C.c_size_t
_cfgp = C.POINTER(GraphConfig)

# name -> (restype, argtypes); must list every function declared in include/ggnn_c.h
class KernelWork(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("points", C.c_uint64), ("n_dist", C.c_uint64),
                ("float_rows", C.c_uint64), ("code_rows", C.c_uint64), ("pops", C.c_uint64),
                ("ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BuildWork(C.Structure):
    _fields_ = [("merge", KernelWork), ("sym", KernelWork)]


SIGNATURES = {
    "ggnn_create": (_int, [C.POINTER(_vp)]),
    "ggnn_destroy": (None, [_vp]),
    "ggnn_last_error": (C.c_char_p, [_vp]),
    "ggnn_version": (C.c_char_p, []),
    "ggnn_set_working_directory": (_int, [_vp, C.c_char_p]),
    "ggnn_set_cpu_memory_limit": (_int, [_vp, _sz]),
    "ggnn_set_reserved_gpu_memory": (_int, [_vp, _sz]),
    "ggnn_set_gpus": (_int, [_vp, C.POINTER(_int), _sz]),
    "ggnn_set_shard_size": (_int, [_vp, _u32]),
    "ggnn_set_return_results_on_gpu": (_int, [_vp, _int]),
    "ggnn_set_base": (_int, [_vp, _vp, _u64, _u32, _int, _int, _int, _int]),
    "ggnn_build": (_int, [_vp, _u32, _f32, _u32, _int]),
    "ggnn_store": (_int, [_vp]),
    "ggnn_load": (_int, [_vp, _u32]),
    "ggnn_query": (_int, [_vp, _vp, _u64, _u32, _int, _int, _int, _u32, _f32, _u32, _int, _vp, _vp,
                          _int]),
    "ggnn_query_async": (_int, [_vp, _vp, _u64, _u32, _int, _int, _u32, _f32, _u32, _int, _vp, _vp,
                                _u32]),
    "ggnn_synchronize": (_int, [_vp]),
    "ggnn_synchronize_slot": (_int, [_vp, _u32]),
    "ggnn_bf_query": (_int, [_vp, _vp, _u64, _u32, _int, _int, _int, _u32, _int, _vp, _vp, _int]),
    "ggnn_get_graph": (_int, [_vp, _u32, C.POINTER(GraphView)]),
    "ggnn_last_timing_ms": (_int, [_vp, C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32)]),
    "ggnn_last_query_counters": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "ggnn_set_collect_counters": (_int, [_vp, _int]),
    "ggnn_set_prescreen": (_int, [_vp, _int]),
    "ggnn_set_build_hooks": (_int, [_vp, _vp, _u64, _int]),
    "ggnn_get_shard_layout": (_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "ggnn_last_query_rows_read": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "ggnn_device_clock_hz": (_int, [_int, C.POINTER(C.c_double)]),
    "ggnn_last_build_work": (_int, [_vp, C.POINTER(BuildWork)]),
    "ggnn_last_query_parts": (_int, [_vp, C.POINTER(_u32)]),
    "ggnn_rccl_ranks": (_int, [_vp, C.POINTER(_u32)]),
    "ggnn_set_hook": (_int, [C.c_char_p, C.c_int64]),
    "ggnn_reset_hook": (_int, [C.c_char_p]),
    "ggnn_get_hook": (_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "ggnn_set_log_level": (None, [_int]),
    "ggnn_graph_config_init": (_int, [_u32, _u32, _u32, _cfgp]),
    "ggnn_query_sizing": (_int, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "ggnn_op_query": (_int, [_vp, _int, _u32, _u32, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u32,
                             _f32, _u32, _int, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "ggnn_prescreen_sizes": (_int, [_u32, _u32, _int, C.POINTER(_u32), C.POINTER(_sz),
                                    C.POINTER(_sz)]),
    "ggnn_op_prescreen_encode": (_int, [_vp, _u32, _u32, _int, _vp, _vp, _vp, _vp]),
    "ggnn_op_prescreen_probe": (_int, [_vp, _vp, _u32, _int, _vp, _u32, _vp, _u32, _vp, _vp, _vp,
                                       _vp]),
    "ggnn_op_query_prescreened": (_int, [_vp, _u32, _u32, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _u32,
                                         _vp, _u32, _f32, _u32, _int, _u32, _u32, _vp, _vp, _vp,
                                         _vp, _vp, _vp]),
    "ggnn_op_bf_query": (_int, [_vp, _int, _u32, _u32, _vp, _u32, _u32, _int, _vp, _vp, _vp]),
    "ggnn_op_bf_query_certified": (_int, [_vp, _int, _u32, _u32, _vp, _u32, _u32, _int, _vp, _vp,
                                          _vp, _vp]),
    "ggnn_last_bf_query_rescanned": (_int, [_vp, C.POINTER(_u32)]),
    "ggnn_last_exchange": (C.c_char_p, [_vp]),
    "ggnn_op_top": (_int, [_vp, _int, _u32, _int, _u32, _vp, _u32, _u32, _u32, _u32, _vp, _vp,
                           _vp]),
    "ggnn_op_merge": (_int, [_vp, _int, _int, _cfgp, _vp, _vp, _vp, _vp, _f32, _u32, _u32, _vp,
                             _vp, _vp, _vp]),
    "ggnn_op_merge_prescreened": (_int, [_vp, _vp, _vp, _int, _cfgp, _vp, _vp, _vp, _vp, _f32, _u32,
                                         _u32, _vp, _vp, _vp, _vp]),
    "ggnn_op_select": (_int, [_cfgp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "ggnn_op_uniform": (_int, [_vp, _u32, _u64, _u64, _vp]),
    "ggnn_op_sym": (_int, [_vp, _int, _int, _u32, _u32, _vp, _vp, _u32, _vp, _f32, _vp, _vp, _u32,
                           _u32, _vp]),
    "ggnn_op_sym_prescreened": (_int, [_vp, _vp, _vp, _int, _u32, _u32, _vp, _vp, _u32, _vp, _f32, _vp,
                                       _vp, _u32, _u32, _vp]),
    "ggnn_op_sym_buffer_merge": (_int, [_u32, _u32, _vp, _vp, _vp, _vp]),
    "ggnn_nn1_stats_scratch_floats": (_sz, []),
    "ggnn_op_nn1_stats": (_int, [_vp, _u32, _vp, _vp, _vp]),
    "ggnn_op_sort_shard_results": (_int, [_u32, _u32, _vp, _vp, _vp]),
    "ggnn_op_merge_results": (_int, [_u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load libggnn_amd.so; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C ggnn_amd/csrc`). ggnn_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def set_hook(name, value):
    """ggnn_set_hook: process-wide test / tuning hook (names in include/ggnn_c.h)"""
    st = lib().ggnn_set_hook(name.encode(), int(value))
    if st != 0:
        raise ValueError(f"unknown hook {name!r}")


def reset_hook(name):
    st = lib().ggnn_reset_hook(name.encode())
    if st != 0:
        raise ValueError(f"unknown hook {name!r}")


def get_hook(name):
    v = C.c_int64()
    if lib().ggnn_get_hook(name.encode(), C.byref(v)) != 0:
        raise ValueError(f"unknown hook {name!r}")
    return int(v.value)


class hooks:
    """with hooks(VIS_SLOTS=1, BF_SLICES=7): ...  -- set for the block, reset afterwards"""

    def __init__(self, **values):
        self.values = values

    def __enter__(self):
        for k, v in self.values.items():
            set_hook(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.values:
            reset_hook(k)
        return False


def check(status, handle=None):
    if status != OK:
        msg = lib().ggnn_last_error(handle)
        raise_for_status(status, msg.decode() if msg else f"ggnn status {status}")
