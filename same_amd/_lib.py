"""ctypes binding of libsame_hip.so (include/same_hip.h: the path's entry points; include/same_hip_diag.h: measurement hooks).

The product path has no CPU fallback: if the shared library is missing, or no MI355X is
visible when a compute entry point is called, this module raises.  Nothing here imports the
oracle.
"""
import ctypes
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAME_HIP_LIB: a measurement hook (an A/B build of the library)
LIB_PATH = os.environ.get("SAME_HIP_LIB") or os.path.join(_HERE, "libsame_hip.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_flt = ctypes.c_float
c_vp = ctypes.c_void_p
c_sz = ctypes.c_size_t

UNIQUE_ID_BYTES = 128
ABI_VERSION = 8
DT_U8, DT_I32, DT_U64, DT_F64 = 0, 1, 2, 3     # SAME_DT_*
OP_SUM, OP_MAX, OP_MIN = 0, 1, 2               # SAME_OP_*
SPREAD_INFO_LEN = 14                           # SAME_SPREAD_INFO_LEN
MAX_KNN = 448
MAX_TYPES = 4096

# name -> argtypes, exactly the declarations of include/same_hip.h and include/same_hip_diag.h (restype int unless noted)
_PROTOTYPES = {
    "same_abi_version": [],
    "same_device_count": [ctypes.POINTER(c_int)],
    "same_ctx_create": [c_int, ctypes.POINTER(c_vp)],
    "same_ctx_destroy": [c_vp],
    "same_ctx_sync": [c_vp],
    "same_strerror": [c_int],
    "same_last_error": [c_vp],
    "same_ctx_info": [c_vp, ctypes.c_char_p, c_sz, ctypes.POINTER(c_int), ctypes.POINTER(c_i64)],
    "same_ctx_pci_bus_id": [c_vp, ctypes.c_char_p, c_sz],
    "same_ctx_stat": [c_vp, c_int, ctypes.POINTER(c_i64)],
    "same_dev_alloc": [c_vp, c_sz, ctypes.POINTER(c_vp)],
    "same_dev_free": [c_vp, c_vp],
    "same_dev_alloc_spread": [c_vp, c_sz, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64)],
    "same_h2d": [c_vp, c_vp, c_vp, c_sz],
    "same_d2h": [c_vp, c_vp, c_vp, c_sz],
    "same_dev_memset": [c_vp, c_vp, c_int, c_sz],
    "same_d2d": [c_vp, c_vp, c_vp, c_sz],
    "same_dev_mem_info": [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)],
    "same_ctx_release_scratch": [c_vp],
    "same_timer_start": [c_vp],
    "same_timer_stop": [c_vp, ctypes.POINTER(c_flt)],
    "same_timer_mark": [c_vp],
    "same_timer_read": [c_vp, ctypes.POINTER(c_flt)],
    "same_pair_cost_f64": [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_dbl, c_vp],
    "same_pair_cost_f32": [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_flt, c_vp],
    "same_dense_cost_f64_dev": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_dbl, c_vp, c_i64],
    "same_dense_cost_f32_dev": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_flt, c_vp, c_i64],
    "same_dense_cost_f64": [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_i64, c_i64, c_dbl, c_vp, c_i64],
    "same_dense_cost_f32": [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_i64, c_i64, c_flt, c_vp, c_i64],
    "same_quantize_u32_dev": [c_vp, c_vp, c_i64, c_dbl, c_dbl, c_vp],
    "same_dense_cost_q32_dev": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_dbl, c_dbl, c_dbl, c_vp, c_i64],
    "same_knn_prune": [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_dbl, c_int, c_vp, c_vp, c_vp],
    "same_knn_prune_dev": [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_dbl, c_int, c_vp, c_vp, c_vp],
    "same_knn_index_build": [c_vp, c_vp, c_i64, c_dbl, ctypes.POINTER(c_vp)],
    "same_knn_index_destroy": [c_vp],
    "same_knn_prune_indexed_dev": [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp],
    "same_padded_cost_f64_dev": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_dbl, c_vp],
    "same_padded_cost_f32_dev": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_int, c_vp, c_flt, c_vp],
    "same_tri_classify": [c_vp, c_vp, c_i64, c_vp, c_i64, c_dbl, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp],
    "same_tri_sign_weight": [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp],
    "same_sweep_bind": [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, ctypes.POINTER(c_vp)],
    "same_sweep_unbind": [c_vp],
    "same_orient_sweep": [c_vp, c_vp, c_i64, ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64), c_vp],
    "same_orient_sweep_x": [c_vp, c_vp, c_i64, ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64), c_vp, c_vp, c_vp],
    "same_xyorder_sweep": [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    "same_area_flip": [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    "same_tri_classify_dev": [c_vp, c_vp, c_vp, c_i64, c_dbl, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp],
    "same_tri_sign_weight_dev": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp],
    "same_area_flip_dev": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    "same_xyorder_sweep_dev": [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    "same_orient_sweep_dev": [c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64)],
    "same_orient_flags_dev": [c_vp, c_vp, c_i64, c_i64, c_vp],
    "same_orient_from_flags_dev": [c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64)],
    "same_first_candidate_dev": [c_vp, c_vp, c_i64, c_int, c_vp],
    "same_pair_rowmin": [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp],
    "same_assign_matrix": [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_dbl, c_vp],
    "same_greedy_match": [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, ctypes.POINTER(c_int)],
    "same_tri_flip_stats": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp],
    "same_collapse_candidates": [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_dbl, c_int, c_dbl, c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp],
    "same_greedy_disjoint": [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, ctypes.POINTER(c_int)],
    "same_batched_assign": [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    "same_eager_signs": [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_vp],
    "same_window_count": [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp],
    "same_section_create": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_int, ctypes.POINTER(c_vp)],
    "same_section_bin": [c_vp, c_dbl, c_dbl, c_dbl, c_dbl],
    "same_section_destroy": [c_vp],
    "same_window_create": [c_vp, ctypes.POINTER(c_vp)],
    "same_window_destroy": [c_vp],
    "same_window_stage": [c_vp, c_int, c_vp, c_vp, c_vp, c_dbl, c_int, c_dbl, c_vp],
    "same_window_fetch": [c_vp, c_int, c_vp, c_i64],
    "same_window_filter_finish": [c_vp, c_int, c_vp, c_vp, c_int, c_dbl, c_int, c_dbl, c_dbl, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp],
    "same_merge_dedup": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(c_i64)],
    "same_delaunay2d": [c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(c_i64), c_dbl, ctypes.POINTER(c_dbl)],
    "same_section_set_codes": [c_vp, c_vp, c_i64],
    "same_merge_acc_create": [c_vp, ctypes.POINTER(c_vp)],
    "same_merge_acc_destroy": [c_vp],
    "same_merge_acc_begin": [c_vp, c_i64, c_int, c_vp, c_vp, c_dbl, c_int],
    "same_window_collect": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp],
    "same_merge_acc_load": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64],
    "same_merge_acc_resolve": [c_vp, c_int, c_vp, c_vp, c_vp],
    "same_merge_acc_finish": [c_vp, c_vp, c_i64, ctypes.POINTER(c_i64)],
    "same_merge_acc_plain": [c_vp, c_int, ctypes.POINTER(c_i64)],
    "same_merge_acc_fetch": [c_vp, c_int, c_vp, c_i64],
    "same_merge_acc_columns": [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_i64],
    "same_host_alloc": [c_vp, c_sz, ctypes.POINTER(c_vp)],
    "same_host_free": [c_vp, c_vp],
    "same_comm_unique_id": [c_vp],
    "same_comm_init": [c_vp, c_int, c_int, c_vp],
    "same_comm_destroy": [c_vp],
    "same_allgather_dev": [c_vp, c_vp, c_vp, c_sz],
    "same_allgather_dev_async": [c_vp, c_vp, c_vp, c_sz],
    "same_comm_wait": [c_vp],
    "same_allreduce_dev": [c_vp, c_vp, c_sz, c_int, c_int],
    "same_comm_group_start": [c_vp],
    "same_comm_group_end": [c_vp],
    "same_comm_info": [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "same_comm_device": [c_vp, ctypes.POINTER(c_int)],
    "same_comm_gather_time": [c_vp, ctypes.POINTER(c_flt), ctypes.POINTER(c_i64)],
}
EXPORTS = tuple(_PROTOTYPES)


SAME_EINVAL, SAME_ENOMEM, SAME_EIO, SAME_ENODEV, SAME_ERANGE, SAME_EUNSURE = -22, -12, -5, -19, -34, -11   # include/same_hip.h


class SameHipError(RuntimeError):
    """A libsame_hip entry point returned a negative code."""

    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__(f"{what}: {detail}" if detail else what)


_lib = None
_lib_lock = threading.Lock()


def load():
    """Load libsame_hip.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SameHipError(-2, f"{LIB_PATH} is missing; build it with `make -C same_amd/csrc` "
                                       "(the product path has no CPU fallback)")
            L = ctypes.CDLL(LIB_PATH)
            for name, argtypes in _PROTOTYPES.items():
                fn = getattr(L, name)
                fn.argtypes = argtypes
                fn.restype = c_int
            L.same_strerror.restype = ctypes.c_char_p
            L.same_last_error.restype = ctypes.c_char_p
            L.same_ctx_destroy.restype = None
            L.same_sweep_unbind.restype = None
            L.same_knn_index_destroy.restype = None
            L.same_section_destroy.restype = None
            L.same_window_destroy.restype = None
            if L.same_abi_version() != ABI_VERSION:
                raise SameHipError(-22, "libsame_hip ABI version mismatch")
            _lib = L
    return _lib


_instrumented = False


def instrument():
    """Account the wall time of every libsame_hip call to `_trace` under "lib:<entry point>" (host-buffer entry points are
    synchronous, so this is device + copy + driver time; what is left of a stage's wall time is Python / pandas / scipy glue).
    bench.py --workload cfg5 uses it for the host-glue share; off unless called."""
    global _instrumented
    import time

    from . import _trace

    L = load()
    with _lib_lock:
        if _instrumented:
            return
        for name in _PROTOTYPES:
            if name in ("same_strerror", "same_last_error", "same_abi_version"):
                continue
            fn = getattr(L, name)

            def timed(*a, _fn=fn, _key="lib:" + name):
                t0 = time.perf_counter()
                try:
                    return _fn(*a)
                finally:
                    _trace.add(_key, time.perf_counter() - t0)

            setattr(L, name, timed)     # instance attribute: shadows the ctypes function pointer for every later lookup
        _instrumented = True


def device_count():
    n = c_int(0)
    load().same_device_count(ctypes.byref(n))
    return n.value


def _ptr(a):
    return None if a is None else a.ctypes.data


def as_c(a, dtype):
    """C-contiguous array of `dtype` (no copy when already so)."""
    return np.ascontiguousarray(a, dtype=dtype)


class DeviceBuffer:
    """A block of HBM owned by a Context (same_dev_alloc)."""

    def __init__(self, ctx, nbytes, spread=False):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = c_vp()
        self.spread_info = None
        if spread:
            info = (c_i64 * SPREAD_INFO_LEN)()
            ctx.check(ctx.lib.same_dev_alloc_spread(ctx.handle, self.nbytes, ctypes.byref(p), info), "same_dev_alloc_spread")
            self.spread_info = {"spread": bool(info[0]), "chunks_gib": int(info[1]),
                                "per_region": [int(info[2]), int(info[3]), int(info[4])],
                                "straddling": int(info[5]), "examined": int(info[6]), "seconds": info[7] * 1e-6,
                                "same_region_level_gbps": int(info[8]), "verified": bool(info[9]), "final_store_gbps": int(info[10]),
                                "pairs_checked": int(info[11]), "pairs_as_labelled": int(info[12]), "stopped_at_time_bound": bool(info[13])}
        else:
            ctx.check(ctx.lib.same_dev_alloc(ctx.handle, self.nbytes, ctypes.byref(p)), "same_dev_alloc")
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.same_h2d(self.ctx.handle, self.ptr, arr.ctypes.data, arr.nbytes), "same_h2d")
        return self

    def download(self, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.same_d2h(self.ctx.handle, out.ctypes.data, self.ptr + offset_bytes, out.nbytes), "same_d2h")
        return out

    def free(self):
        if self.ptr is not None and self.ctx.handle:
            self.ctx.lib.same_dev_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU, one stream (same_ctx).  Not shared between concurrent callers."""

    def __init__(self, device=0):
        self.lib = load()
        h = c_vp()
        rc = self.lib.same_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise SameHipError(rc, f"same_ctx_create(device={device})", self.lib.same_strerror(rc).decode())
        self.handle = h.value
        self.device = int(device)
        self.lock = threading.RLock()

    def check(self, rc, what):
        if rc != 0:
            detail = self.lib.same_last_error(self.handle).decode() if self.handle else ""
            raise SameHipError(rc, f"{what}: {self.lib.same_strerror(rc).decode()}", detail)

    def info(self):
        name = ctypes.create_string_buffer(128)
        cu, hbm = c_int(0), c_i64(0)
        self.check(self.lib.same_ctx_info(self.handle, name, 128, ctypes.byref(cu), ctypes.byref(hbm)), "same_ctx_info")
        return {"arch": name.value.decode(), "cu_count": cu.value, "hbm_bytes": hbm.value}

    def pci_bus_id(self):
        buf = ctypes.create_string_buffer(64)
        self.check(self.lib.same_ctx_pci_bus_id(self.handle, buf, 64), "same_ctx_pci_bus_id")
        return buf.value.decode().lower()

    STAT_NAMES = ("launches", "fills", "copies", "waits", "greedy_readbacks")       # SAME_STAT_*

    def stats(self):
        """What the library has asked of the HIP runtime on this context so far, for the entry points that count (the window path,
        the greedy start): {'launches', 'fills', 'copies', 'waits', 'greedy_readbacks'} (same_ctx_stat)."""
        out, v = {}, c_i64(0)
        for which, name in enumerate(self.STAT_NAMES):
            self.check(self.lib.same_ctx_stat(self.handle, which, ctypes.byref(v)), "same_ctx_stat")
            out[name] = v.value
        return out

    def mem_info(self):
        """(free bytes, total bytes) of the card right now."""
        f, t = c_i64(0), c_i64(0)
        self.check(self.lib.same_dev_mem_info(self.handle, ctypes.byref(f), ctypes.byref(t)), "same_dev_mem_info")
        return f.value, t.value

    def mem_free(self):
        return self.mem_info()[0]

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def alloc_spread(self, nbytes):
        """A large streaming-output buffer laid over the card's HBM regions (same_dev_alloc_spread); `.spread_info` says how."""
        return DeviceBuffer(self, nbytes, spread=True)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return self.alloc(max(arr.nbytes, 16)).upload(arr)

    def sync(self):
        self.check(self.lib.same_ctx_sync(self.handle), "same_ctx_sync")

    def release_scratch(self):
        """Free the staging blocks the host-buffer entry points grew (they are otherwise kept for reuse)."""
        self.check(self.lib.same_ctx_release_scratch(self.handle), "same_ctx_release_scratch")

    def timer_start(self):
        self.check(self.lib.same_timer_start(self.handle), "same_timer_start")

    def timer_stop(self):
        ms = c_flt(0)
        self.check(self.lib.same_timer_stop(self.handle, ctypes.byref(ms)), "same_timer_stop")
        return ms.value

    def close(self):
        for state in self.__dict__.pop("_device_windows", []):      # window states cached by windows.iter_device_windows
            state.close()
        if self.handle and "same_amd.windows" in sys.modules:       # page-locked blocks of this context waiting in the table pool
            sys.modules["same_amd.windows"].PINNED_BLOCKS.drop(self)
        if self.handle:
            self.lib.same_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_n_devices = None
_default_lock = threading.Lock()


def default_context(device=None):
    """Process-wide context for `device` (default: $SAME_HIP_DEVICE or LOCAL_RANK or 0)."""
    if device is None:
        global _n_devices
        device = int(os.environ.get("SAME_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if _n_devices is None or _n_devices <= 0:
            _n_devices = device_count()      # asked once: hipGetDeviceCount costs ~0.15 ms and every op of a window resolves its context
        if _n_devices > 0:
            device %= _n_devices
    with _default_lock:
        ctx = _default_ctx.get(device)
        if ctx is None or not ctx.handle:
            ctx = _default_ctx[device] = Context(device)
    return ctx
