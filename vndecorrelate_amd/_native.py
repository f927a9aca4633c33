"""ctypes binding of ``libvnd_amd.so`` (C ABI in ``include/vnd_amd.h``).

This is the only way the package computes anything: there is no NumPy or CPU
fallback.  If the shared library is missing, or no gfx950 device is visible,
the first call that needs the GPU raises ``RuntimeError`` - loudly, by design.
"""
from __future__ import annotations

import ctypes
import os
import pathlib
import threading
import weakref
from typing import Optional

import numpy as np

ABI_VERSION = 2
MODE_EXACT = 0   # acc = f32(acc + f32(x*w)) : bit-identical to the reference's NumPy paths
MODE_FMA = 1     # acc = fma(x, w, acc), table order
MODE_FAST = 2    # fma, free summation order, gains folded into weights: the throughput mode
MOMENTS = 8      # VND_MOMENTS: doubles per candidate returned by the scan
MAX_STREAMS_PER_CALL = 65535   # VND_MAX_STREAMS: the decorrelate / Haas kernels index streams by gridDim.y
NORMALIZE_OFF, NORMALIZE_RMS, NORMALIZE_RMS_REFERENCE_ORDER = 0, 1, 2   # the `normalize` argument of the decorrelate calls

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get('VND_AMD_LIBRARY', _PKG / 'libvnd_amd.so'))   # override: tuning builds only

_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_f32p = ctypes.POINTER(ctypes.c_float)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)

# name -> (restype, argtypes); also the list tests/test_abi.py checks against the header
SIGNATURES = {
    'vnd_abi_version': (ctypes.c_int, []),
    'vnd_last_error': (ctypes.c_char_p, []),
    'vnd_device_count': (ctypes.c_int, [_c_i32p]),
    'vnd_ctx_create': (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    'vnd_ctx_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'vnd_ctx_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32, _c_i32p,
                                    ctypes.POINTER(ctypes.c_int64), _c_i32p]),
    'vnd_taps_create': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _c_i32p, _c_i32p, _c_f32p,
                                       _c_i32p, _c_i32p, _c_f32p, _c_u8p, ctypes.c_int32,
                                       ctypes.POINTER(ctypes.c_void_p)]),
    'vnd_taps_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'vnd_taps_info': (ctypes.c_int, [ctypes.c_void_p, _c_i32p, _c_i32p, _c_i32p]),
    'vnd_taps_serialize': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.POINTER(ctypes.c_int64)]),
    'vnd_taps_deserialize': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.POINTER(ctypes.c_void_p)]),
    'vnd_shard_range': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'vnd_taps_broadcast_rccl': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                               ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    'vnd_convolve_f32_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    'vnd_convolve_f32_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _c_f32p,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_int32]),
    'vnd_decorrelate_workspace_bytes': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                       ctypes.POINTER(ctypes.c_int64)]),
    'vnd_decorrelate_f32_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
                                               ctypes.c_float, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    'vnd_decorrelate_f32_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _c_f32p,
                                                ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
                                                ctypes.c_float]),
    'vnd_convolve_fanout_f32_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    'vnd_convolve_fanout_f32_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _c_f32p,
                                                    ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                    ctypes.c_int32]),
    'vnd_decorrelate_fanout_f32_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                      ctypes.c_int32, ctypes.c_double, ctypes.c_int32,
                                                      ctypes.c_float, ctypes.c_void_p, ctypes.c_int64,
                                                      ctypes.c_void_p]),
    'vnd_decorrelate_fanout_f32_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _c_f32p,
                                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                       ctypes.c_double, ctypes.c_int32, ctypes.c_float]),
    'vnd_describe_fanout_launch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                  ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_char_p, ctypes.c_int32]),
    'vnd_polar_moments_workspace_bytes': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int32,
                                                         ctypes.POINTER(ctypes.c_int64)]),
    'vnd_polar_moments_f32_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    'vnd_scan_bank_f32_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_f32p, ctypes.c_int64,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]),
    'vnd_haas_f64_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p]),
    'vnd_haas_f64_host': (ctypes.c_int, [ctypes.c_void_p, _c_f32p, ctypes.POINTER(ctypes.c_double), ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_double]),
    'vnd_convolve_promote_host': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _c_i32p, _c_i32p,
                                                 ctypes.POINTER(ctypes.c_double), ctypes.c_void_p, ctypes.c_int32,
                                                 _c_f32p, ctypes.c_int64, ctypes.c_int64]),
    'vnd_host_alloc': (ctypes.c_int, [ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    'vnd_host_buffers_mapped': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.POINTER(ctypes.c_int32)]),
    'vnd_host_free': (ctypes.c_int, [ctypes.c_void_p]),
    'vnd_prepare_launch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int32]),
    'vnd_describe_launch': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_char_p, ctypes.c_int32]),
}

# include/vnd_amd_internal.h: measurement, tuning and diagnosis hooks (bench.py, tools/, tests) - not the drop-in ABI
INTERNAL_SIGNATURES = {
    'vnd_time_convolve_f32_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                                 _c_f32p]),
    'vnd_time_copy_f32_dev': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    'vnd_code_object_private_bytes': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    'vnd_spec_kernel_source': (ctypes.c_int, [ctypes.c_int32, _c_i32p, _c_i32p, _c_f32p, ctypes.c_int32, ctypes.c_char_p,
                                              ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    'vnd_window_kernel_source': (ctypes.c_int, [ctypes.c_int32, _c_i32p, _c_i32p, _c_f32p, _c_i32p, _c_i32p, _c_f32p,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64,
                                                ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                                ctypes.POINTER(ctypes.c_int64)]),
    'vnd_set_variant': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    'vnd_tuning_read': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    'vnd_debug_read_stamps': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
}

_lib = None
_lib_lock = threading.Lock()


def _preload_hip_runtime() -> None:
    """Keep ONE HIP runtime in the process.

    PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (soname
    ``libamdhip64.so.7``) and ask for it by the unversioned name, so if this
    extension pulled in ``/opt/rocm``'s copy first, a later ``import torch`` would
    map a second runtime and see no GPUs.  When torch is installed but not yet
    imported, map its runtime first: our NEEDED ``libamdhip64.so.7`` then binds
    to it by soname, and torch later finds the same file already loaded.
    """
    try:
        with open('/proc/self/maps') as maps:
            if any('libamdhip64' in line for line in maps):
                return
    except OSError:
        pass
    import importlib.util
    spec = importlib.util.find_spec('torch')
    if spec is None or not spec.submodule_search_locations:
        return
    cand = pathlib.Path(list(spec.submodule_search_locations)[0]) / 'lib' / 'libamdhip64.so'
    if cand.exists():
        ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)


class NativeError(RuntimeError):
    """The HIP extension is missing or a device call failed."""


def load_library():
    """dlopen the in-tree extension and declare every prototype of the header."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise NativeError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; '
                f'g.build()"` (hipcc --offload-arch=gfx950).  vndecorrelate_amd has no CPU fallback.')
        _preload_hip_runtime()
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in list(SIGNATURES.items()) + list(INTERNAL_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = lib.vnd_abi_version()
        if got != ABI_VERSION:
            raise NativeError(f'{LIB_PATH} has ABI {got}, expected {ABI_VERSION}: rebuild it')
        _lib = lib
        return lib


def shard_range(total: int, world_size: int, rank: int):
    """``vnd_shard_range``: ``(first, count)`` of ``rank``'s contiguous block of ``total`` streams."""
    first, count = ctypes.c_int64(), ctypes.c_int64()
    _check(load_library().vnd_shard_range(total, world_size, rank, ctypes.byref(first), ctypes.byref(count)),
           'vnd_shard_range')
    return first.value, count.value


def _check(rc: int, what: str):
    if rc == 0:
        return
    msg = load_library().vnd_last_error().decode(errors='replace')
    if rc == 1:
        raise ValueError(f'{what}: {msg}')
    raise NativeError(f'{what} failed (status {rc}): {msg}')


def _ptr(a: Optional[np.ndarray], ctype):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctype))


class _PinnedPool:
    """Result arrays of the host API in page-locked memory (``vnd_host_alloc``).

    The reference allocates its result afresh in every call (decorrelation.py:647); a fresh pageable
    array of hundreds of MB costs tens of milliseconds of page faults and a staged download.  Blocks
    are recycled when the NumPy array (and every view of it) is garbage-collected; the pool keeps at
    most ``VND_PINNED_POOL_MB`` (default 2048; 0 turns it off) of idle blocks.  Results below 1 MiB stay
    ordinary NumPy arrays."""

    MIN_BYTES = 1 << 20

    def __init__(self):
        self._lock = threading.Lock()
        self._free: dict = {}            # rounded size -> [ptr, ...]
        self._idle = 0
        self.limit = int(os.environ.get('VND_PINNED_POOL_MB', '2048')) << 20
        self.hits = self.misses = 0

    @staticmethod
    def _round(nbytes: int) -> int:
        size = 1 << 20
        while size < nbytes:
            size <<= 1
        return size if size - nbytes <= nbytes // 4 else ((nbytes + (1 << 20) - 1) >> 20) << 20

    def empty(self, shape, dtype=np.float32) -> np.ndarray:
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if self.limit <= 0 or nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype)
        size = self._round(nbytes)
        ptr = None
        with self._lock:
            blocks = self._free.get(size)
            if blocks:
                ptr = blocks.pop()
                self._idle -= size
                self.hits += 1
        if ptr is None:
            h = ctypes.c_void_p()
            try:
                rc = load_library().vnd_host_alloc(size, ctypes.byref(h))
            except Exception:
                rc = 1
            if rc != 0 or not h.value:
                return np.empty(shape, dtype)          # no pinned memory to be had: an ordinary array
            ptr = h.value
            self.misses += 1
        buf = (ctypes.c_char * nbytes).from_address(ptr)
        weakref.finalize(buf, self._release, ptr, size)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def _release(self, ptr: int, size: int):
        with self._lock:
            if self._idle + size <= self.limit:
                self._free.setdefault(size, []).append(ptr)
                self._idle += size
                return
        try:
            load_library().vnd_host_free(ctypes.c_void_p(ptr))
        except Exception:
            pass

    def trim(self):
        with self._lock:
            blocks = [p for v in self._free.values() for p in v]
            self._free.clear()
            self._idle = 0
        for p in blocks:
            load_library().vnd_host_free(ctypes.c_void_p(p))


pinned_pool = _PinnedPool()


class Context:
    """One per (process, device): owns the stream and staging buffers of the
    synchronous host-pointer calls.  Thread-safe: the library holds a per-context
    mutex across every ``*_host`` entry point (ctypes drops the GIL during the call),
    so concurrent callers are serialised, never interleaved; use one Context per
    thread for host calls that should overlap."""

    def __init__(self, device: int = 0):
        lib = load_library()
        h = ctypes.c_void_p()
        _check(lib.vnd_ctx_create(int(device), ctypes.byref(h)), 'vnd_ctx_create')
        self._lib = lib
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.vnd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise NativeError('context is closed')
        return self._h

    def info(self) -> dict:
        name = ctypes.create_string_buffer(256)
        cus, lds = ctypes.c_int32(), ctypes.c_int32()
        hbm = ctypes.c_int64()
        _check(self._lib.vnd_ctx_info(self.handle, name, 256, ctypes.byref(cus), ctypes.byref(hbm),
                                      ctypes.byref(lds)), 'vnd_ctx_info')
        return {'name': name.value.decode(), 'compute_units': cus.value, 'hbm_bytes': hbm.value,
                'lds_bytes': lds.value}

    def set_variant(self, variant: int):
        _check(self._lib.vnd_set_variant(self.handle, int(variant)), 'vnd_set_variant')

    def time_copy(self, x_ptr: int, y_ptr: int, elems: int, iters: int = 10, stream: int = 0) -> float:
        """Average kernel milliseconds of a plain streaming copy of ``elems`` floats (``vnd_time_copy_f32_dev``:
        the box's streaming ceiling for bench.py; not on the data path)."""
        ms = ctypes.c_float()
        _check(self._lib.vnd_time_copy_f32_dev(self.handle, ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), int(elems),
                                               int(iters), ctypes.c_void_p(stream), ctypes.byref(ms)), 'vnd_time_copy_f32_dev')
        return float(ms.value)


class TapTable:
    """Device-resident, immutable tap table (``vnd_taps``)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self._lib = ctx._lib
        self._h = handle
        c, t, m = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _check(self._lib.vnd_taps_info(handle, ctypes.byref(c), ctypes.byref(t), ctypes.byref(m)),
               'vnd_taps_info')
        self.num_channels, self.total_taps, self.max_index = c.value, t.value, m.value

    @classmethod
    def create(cls, ctx: Context, tap_offsets, tap_index, tap_weight, *, seg_offsets=None,
               seg_end=None, seg_gain=None, chan_flags=None, apply_gain=False) -> 'TapTable':
        tap_offsets = np.ascontiguousarray(tap_offsets, np.int32)
        tap_index = np.ascontiguousarray(tap_index, np.int32)
        tap_weight = np.ascontiguousarray(tap_weight, np.float32)
        channels = len(tap_offsets) - 1
        if seg_offsets is not None:
            seg_offsets = np.ascontiguousarray(seg_offsets, np.int32)
            seg_end = np.ascontiguousarray(seg_end, np.int32)
            seg_gain = np.ascontiguousarray(seg_gain, np.float32)
        if chan_flags is not None:
            chan_flags = np.ascontiguousarray(chan_flags, np.uint8)
        h = ctypes.c_void_p()
        _check(ctx._lib.vnd_taps_create(
            ctx.handle, channels, _ptr(tap_offsets, ctypes.c_int32), _ptr(tap_index, ctypes.c_int32),
            _ptr(tap_weight, ctypes.c_float), _ptr(seg_offsets, ctypes.c_int32),
            _ptr(seg_end, ctypes.c_int32), _ptr(seg_gain, ctypes.c_float),
            _ptr(chan_flags, ctypes.c_uint8), int(bool(apply_gain)), ctypes.byref(h)),
            'vnd_taps_create')
        return cls(ctx, h)

    @classmethod
    def from_bytes(cls, ctx: Context, image: bytes) -> 'TapTable':
        buf = ctypes.create_string_buffer(image, len(image))
        h = ctypes.c_void_p()
        _check(ctx._lib.vnd_taps_deserialize(ctx.handle, buf, len(image), ctypes.byref(h)),
               'vnd_taps_deserialize')
        return cls(ctx, h)

    @classmethod
    def broadcast_rccl(cls, ctx: 'Context', table: Optional['TapTable'], root: int, rank: int, comm: int,
                       stream: int = 0) -> 'TapTable':
        """The C ABI's own table broadcast over an RCCL communicator (``ncclComm_t`` as an integer) - for
        hosts that shard without torch.distributed.  ``table`` is needed on ``root`` only; every rank
        returns a table on its device (the root its own)."""
        h = ctypes.c_void_p((table._h.value if isinstance(table._h, ctypes.c_void_p) else table._h) if (table is not None and rank == root) else None)
        _check(ctx._lib.vnd_taps_broadcast_rccl(ctx.handle, ctypes.byref(h), root, rank, ctypes.c_void_p(comm),
                                                ctypes.c_void_p(stream)), 'vnd_taps_broadcast_rccl')
        return table if rank == root else cls(ctx, h)

    def to_bytes(self) -> bytes:
        need = ctypes.c_int64()
        _check(self._lib.vnd_taps_serialize(self.handle, None, 0, ctypes.byref(need)),
               'vnd_taps_serialize')
        buf = ctypes.create_string_buffer(need.value)
        _check(self._lib.vnd_taps_serialize(self.handle, buf, need.value, ctypes.byref(need)),
               'vnd_taps_serialize')
        return buf.raw[:need.value]

    @property
    def handle(self):
        if not self._h:
            raise NativeError('tap table is closed')
        return self._h

    def close(self):
        if getattr(self, '_h', None):
            self._lib.vnd_taps_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the hot path ---------------------------------------------------------
    # A signal with fewer channels than the table is fanned out: output channel c reads
    # input channel c % in_channels (vnd_*_fanout_*; mono -> stereo, or one signal through
    # a bank of filters).  The result always has the table's channel count.
    def _host_shapes(self, x: np.ndarray, what: str, out: Optional[np.ndarray] = None):
        if x.dtype != np.float32 or not x.flags.c_contiguous:
            raise ValueError(f'{what} wants a C-contiguous float32 array')
        if x.ndim == 2:
            batch, (n, c) = 1, x.shape
        elif x.ndim == 3:
            batch, n, c = x.shape
        else:
            raise ValueError(f'expected (n, C) or (batch, n, C), got {x.shape}')
        shape = x.shape[:-1] + (self.num_channels,)
        if out is None:
            y = pinned_pool.empty(shape, np.float32)
        else:                                   # the caller's block of a larger result (multi.DevicePool): written in place
            if out.dtype != np.float32 or not out.flags.c_contiguous or out.shape != shape:
                raise ValueError(f'{what}: out must be a C-contiguous float32 array of shape {shape}')
            y = out
        return batch, n, c, y

    def convolve_host(self, x: np.ndarray, mode: int = MODE_EXACT, *, out: Optional[np.ndarray] = None) -> np.ndarray:
        """x: C-contiguous float32 ``(n, C)`` or ``(batch, n, C)``; returns a new array (or ``out``, filled)."""
        batch, n, c, y = self._host_shapes(x, 'convolve_host', out)
        if c == self.num_channels:
            _check(self._lib.vnd_convolve_f32_host(self.ctx.handle, self.handle,
                                                   _ptr(x, ctypes.c_float), _ptr(y, ctypes.c_float),
                                                   batch, n, c, int(mode)), 'vnd_convolve_f32_host')
        else:
            _check(self._lib.vnd_convolve_fanout_f32_host(self.ctx.handle, self.handle,
                                                          _ptr(x, ctypes.c_float), _ptr(y, ctypes.c_float),
                                                          batch, n, c, int(mode)), 'vnd_convolve_fanout_f32_host')
        return y

    def decorrelate_host(self, x: np.ndarray, mode: int = MODE_EXACT, *, ms_encode: bool, width,
                         normalize, eps: float = 1e-10, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Convolution + decorrelate epilogue on the device; x as in ``convolve_host``.
        ``normalize``: False/True or one of the ``NORMALIZE_*`` values."""
        batch, n, c, y = self._host_shapes(x, 'decorrelate_host', out)
        if batch > MAX_STREAMS_PER_CALL:
            for first in range(0, batch, MAX_STREAMS_PER_CALL):
                self.decorrelate_host(x[first:first + MAX_STREAMS_PER_CALL], mode, ms_encode=ms_encode, width=width,
                                      normalize=normalize, eps=eps, out=y[first:first + MAX_STREAMS_PER_CALL])
            return y
        tail = (int(mode), int(bool(ms_encode)), int(width is not None), float(width or 0.0),
                int(normalize), float(eps))
        if c == self.num_channels:
            _check(self._lib.vnd_decorrelate_f32_host(
                self.ctx.handle, self.handle, _ptr(x, ctypes.c_float), _ptr(y, ctypes.c_float), batch, n, c,
                *tail), 'vnd_decorrelate_f32_host')
        else:
            _check(self._lib.vnd_decorrelate_fanout_f32_host(
                self.ctx.handle, self.handle, _ptr(x, ctypes.c_float), _ptr(y, ctypes.c_float), batch, n, c,
                *tail), 'vnd_decorrelate_fanout_f32_host')
        return y

    def decorrelate_device(self, x_ptr: int, y_ptr: int, batch: int, n: int, channels: int, *, mode: int,
                           ms_encode: bool, width, normalize: bool, workspace_ptr: int, workspace_bytes: int,
                           eps: float = 1e-10, stream: int = 0):
        """``channels`` = channels of x; y has the table's channel count."""
        fn, name = ((self._lib.vnd_decorrelate_f32_dev, 'vnd_decorrelate_f32_dev')
                    if channels == self.num_channels else
                    (self._lib.vnd_decorrelate_fanout_f32_dev, 'vnd_decorrelate_fanout_f32_dev'))
        _check(fn(self.ctx.handle, self.handle, ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), batch, n, channels,
                  int(mode), int(bool(ms_encode)), int(width is not None), float(width or 0.0), int(normalize),
                  float(eps), ctypes.c_void_p(workspace_ptr), workspace_bytes, ctypes.c_void_p(stream)), name)

    def convolve_device(self, x_ptr: int, y_ptr: int, batch: int, n: int, channels: int,
                        mode: int = MODE_EXACT, stream: int = 0):
        """Enqueue on ``stream`` (a hipStream_t as int); pointers are device addresses.
        ``channels`` = channels of x; y has the table's channel count."""
        fn, name = ((self._lib.vnd_convolve_f32_dev, 'vnd_convolve_f32_dev')
                    if channels == self.num_channels else
                    (self._lib.vnd_convolve_fanout_f32_dev, 'vnd_convolve_fanout_f32_dev'))
        _check(fn(self.ctx.handle, self.handle, ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr),
                  batch, n, channels, int(mode), ctypes.c_void_p(stream)), name)

    def scan_host(self, x: np.ndarray, mode: int = MODE_EXACT) -> np.ndarray:
        """Candidate scan: this table is a bank of F stereo pairs, ``x`` a ``(n, 1)`` or
        ``(n, 2)`` float32 signal; returns ``(F, 8)`` float64 polar moments (``vnd_amd.h``)."""
        if x.dtype != np.float32 or not x.flags.c_contiguous or x.ndim != 2:
            raise ValueError('scan_host wants a C-contiguous float32 (n, channels) array')
        out = np.zeros((self.num_channels // 2, MOMENTS), np.float64)
        _check(self._lib.vnd_scan_bank_f32_host(self.ctx.handle, self.handle, _ptr(x, ctypes.c_float), x.shape[0],
                                                x.shape[1], int(mode), _ptr(out, ctypes.c_double)),
               'vnd_scan_bank_f32_host')
        return out

    def time_device(self, x_ptr: int, y_ptr: int, batch: int, n: int, channels: int, *, mode: int,
                    n_buffers: int, stride_elems: int, iters: int, stream: int = 0) -> float:
        """Average milliseconds per launch between two hipEvents on ``stream``."""
        ms = ctypes.c_float()
        _check(self._lib.vnd_time_convolve_f32_dev(
            self.ctx.handle, self.handle, ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), batch, n,
            channels, int(mode), n_buffers, stride_elems, iters, ctypes.c_void_p(stream),
            ctypes.byref(ms)), 'vnd_time_convolve_f32_dev')
        return ms.value

    def prepare(self, batch: int, n: int, channels: int, mode: int = MODE_EXACT) -> None:
        """Build now the per-table kernel that launches of this shape would use (``vnd_prepare_launch``): small
        launches never trigger a build themselves, so a host that repeats one small shape prepares it once."""
        _check(self._lib.vnd_prepare_launch(self.ctx.handle, self.handle, batch, n, channels, int(mode)), 'vnd_prepare_launch')

    def describe(self, batch: int, n: int, channels: int, mode: int = MODE_EXACT) -> str:
        buf = ctypes.create_string_buffer(512)
        fn, name = ((self._lib.vnd_describe_launch, 'vnd_describe_launch')
                    if channels == self.num_channels else
                    (self._lib.vnd_describe_fanout_launch, 'vnd_describe_fanout_launch'))
        _check(fn(self.ctx.handle, self.handle, batch, n, channels, int(mode), buf, 512), name)
        return buf.value.decode()

    def read_stamps(self, batch: int, n: int, channels: int, mode: int = MODE_EXACT) -> np.ndarray:
        """Phase stamps of the window-form kernel of this launch shape (``vnd_debug_read_stamps``; diagnosis builds
        under ``VND_TUNING=1 VND_WIN_STAMPS=<workgroups>``): ``(workgroups, 16)`` uint64, empty without such a build."""
        count = ctypes.c_int64()
        _check(self._lib.vnd_debug_read_stamps(self.ctx.handle, self.handle, batch, n, channels, int(mode), None, 0,
                                               ctypes.byref(count)), 'vnd_debug_read_stamps')
        out = np.zeros(count.value, np.uint64)
        if count.value:
            _check(self._lib.vnd_debug_read_stamps(self.ctx.handle, self.handle, batch, n, channels, int(mode),
                                                   out.ctypes.data_as(ctypes.c_void_p), count.value, ctypes.byref(count)),
                   'vnd_debug_read_stamps')
        return out.reshape(-1, 16)


def decorrelate_workspace_bytes(batch: int, n: int, channels: int) -> int:
    need = ctypes.c_int64()
    _check(load_library().vnd_decorrelate_workspace_bytes(batch, n, channels, ctypes.byref(need)),
           'vnd_decorrelate_workspace_bytes')
    return need.value


def convolve_promote_host(ctx: 'Context', x: np.ndarray, tap_offsets: np.ndarray, tap_index: np.ndarray,
                          tap_weight: np.ndarray) -> np.ndarray:
    """The function path on operands NumPy promotes to float64 (``vnd_convolve_promote_host``):
    x float32 or float64 ``(n, C)`` / ``(batch, n, C)``, float64 weights; float32 result."""
    if x.dtype not in (np.float32, np.float64) or not x.flags.c_contiguous or x.ndim not in (2, 3):
        raise ValueError('convolve_promote_host wants a C-contiguous float32/float64 (n, C) or (batch, n, C) array')
    batch = 1 if x.ndim == 2 else x.shape[0]
    n, c = x.shape[-2:]
    offs = np.ascontiguousarray(tap_offsets, np.int32)
    idx = np.ascontiguousarray(tap_index, np.int32)
    w = np.ascontiguousarray(tap_weight, np.float64)
    y = np.empty(x.shape, np.float32)
    _check(ctx._lib.vnd_convolve_promote_host(ctx.handle, c, _ptr(offs, ctypes.c_int32), _ptr(idx, ctypes.c_int32),
                                              _ptr(w, ctypes.c_double), ctypes.c_void_p(x.ctypes.data),
                                              int(x.dtype == np.float64), _ptr(y, ctypes.c_float), batch, n),
           'vnd_convolve_promote_host')
    return y


def haas_host(ctx: 'Context', x: np.ndarray, *, delay: int, delayed_channel: int, ms_mode: bool, width) -> np.ndarray:
    """HaasEffect on the device from host memory: x float32 ``(n, 1|2)`` or ``(batch, n, 1|2)``;
    returns float64 ``(..., n + delay, 2)``."""
    if x.dtype != np.float32 or not x.flags.c_contiguous or x.ndim not in (2, 3):
        raise ValueError('haas_host wants a C-contiguous float32 (n, C) or (batch, n, C) array')
    batch = 1 if x.ndim == 2 else x.shape[0]
    n, c = x.shape[-2:]
    y = np.empty(x.shape[:-2] + (n + int(delay), 2), np.float64)
    _check(ctx._lib.vnd_haas_f64_host(ctx.handle, _ptr(x, ctypes.c_float), _ptr(y, ctypes.c_double), batch, n, c,
                                      int(delay), int(delayed_channel), int(bool(ms_mode)), int(width is not None),
                                      float(width or 0.0)), 'vnd_haas_f64_host')
    return y


def haas_device(ctx: 'Context', x_ptr: int, y_ptr: int, batch: int, n: int, channels: int, *, delay: int,
                delayed_channel: int, ms_mode: bool, width, stream: int = 0):
    _check(ctx._lib.vnd_haas_f64_dev(ctx.handle, ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), batch, n, channels,
                                     int(delay), int(delayed_channel), int(bool(ms_mode)), int(width is not None),
                                     float(width or 0.0), ctypes.c_void_p(stream)), 'vnd_haas_f64_dev')


def polar_moments_workspace_bytes(n: int, pairs: int) -> int:
    need = ctypes.c_int64()
    _check(load_library().vnd_polar_moments_workspace_bytes(n, pairs, ctypes.byref(need)),
           'vnd_polar_moments_workspace_bytes')
    return need.value


def polar_moments_device(ctx: 'Context', y_ptr: int, n: int, pairs: int, moments_ptr: int, workspace_ptr: int,
                         workspace_bytes: int, stream: int = 0):
    """Reduce a device array ``y[n][2*pairs]`` to ``moments[pairs][8]`` (device doubles) on ``stream``."""
    _check(ctx._lib.vnd_polar_moments_f32_dev(ctx.handle, ctypes.c_void_p(y_ptr), n, pairs,
                                              ctypes.c_void_p(moments_ptr), ctypes.c_void_p(workspace_ptr),
                                              workspace_bytes, ctypes.c_void_p(stream)),
           'vnd_polar_moments_f32_dev')


def spec_kernel_source(tap_offsets, tap_index, tap_weight, mode: int = MODE_FAST) -> str:
    """HIP source of the per-table fast kernel the library would compile with hipRTC
    (``vnd_spec_kernel_source``; needs no device)."""
    offs = np.ascontiguousarray(tap_offsets, np.int32)
    idx = np.ascontiguousarray(tap_index, np.int32)
    w = np.ascontiguousarray(tap_weight, np.float32)
    lib = load_library()
    need = ctypes.c_int64()
    args = (len(offs) - 1, _ptr(offs, ctypes.c_int32), _ptr(idx, ctypes.c_int32), _ptr(w, ctypes.c_float), int(mode))
    _check(lib.vnd_spec_kernel_source(*args, None, 0, ctypes.byref(need)), 'vnd_spec_kernel_source')
    buf = ctypes.create_string_buffer(need.value)
    _check(lib.vnd_spec_kernel_source(*args, buf, need.value, ctypes.byref(need)), 'vnd_spec_kernel_source')
    return buf.value.decode()


def window_kernel_source(tap_offsets, tap_index, tap_weight, mode: int = MODE_FAST, frames_per_lane: int = 32,
                         threads: int = 256, with_traffic: bool = False, *, seg_offsets=None, seg_end=None,
                         seg_gain=None, apply_gain: bool = False):
    """HIP source of the WINDOW form of the per-table kernel (``vnd_window_kernel_source``; needs no device).
    ``seg_*``: a class-path table as for ``TapTable.create``.  ``with_traffic``: also return (LDS bytes one lane
    reads per tile, (tap, output) products they feed)."""
    offs = np.ascontiguousarray(tap_offsets, np.int32)
    idx = np.ascontiguousarray(tap_index, np.int32)
    w = np.ascontiguousarray(tap_weight, np.float32)
    so = se = sg = None
    if seg_offsets is not None:
        so, se = np.ascontiguousarray(seg_offsets, np.int32), np.ascontiguousarray(seg_end, np.int32)
        sg = np.ascontiguousarray(seg_gain, np.float32)
    lib = load_library()
    need, lb, fm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    args = (len(offs) - 1, _ptr(offs, ctypes.c_int32), _ptr(idx, ctypes.c_int32), _ptr(w, ctypes.c_float),
            _ptr(so, ctypes.c_int32), _ptr(se, ctypes.c_int32), _ptr(sg, ctypes.c_float), int(bool(apply_gain)), int(mode),
            int(frames_per_lane), int(threads))
    _check(lib.vnd_window_kernel_source(*args, None, 0, ctypes.byref(need), ctypes.byref(lb), ctypes.byref(fm)),
           'vnd_window_kernel_source')
    buf = ctypes.create_string_buffer(need.value)
    _check(lib.vnd_window_kernel_source(*args, buf, need.value, ctypes.byref(need), None, None), 'vnd_window_kernel_source')
    src = buf.value.decode()
    return (src, lb.value, fm.value) if with_traffic else src


def code_object_private_bytes(image: bytes, kernel: str) -> int:
    """Private (scratch) bytes per lane of ``kernel`` in a gfx950 code object (``vnd_code_object_private_bytes``);
    -1 if the image has no such kernel."""
    out = ctypes.c_int64()
    _check(load_library().vnd_code_object_private_bytes(image, len(image), kernel.encode(), ctypes.byref(out)),
           'vnd_code_object_private_bytes')
    return out.value


def host_buffers_mapped(x: np.ndarray, y: np.ndarray) -> bool:
    """True if a ``*_host`` convolution from ``x`` into ``y`` would run in place on the two buffers (both
    page-locked and mapped: ``vnd_host_buffers_mapped``)."""
    flag = ctypes.c_int32()
    _check(load_library().vnd_host_buffers_mapped(x.ctypes.data, x.nbytes, y.ctypes.data, y.nbytes, ctypes.byref(flag)),
           'vnd_host_buffers_mapped')
    return bool(flag.value)


def device_count() -> int:
    n = ctypes.c_int32()
    rc = load_library().vnd_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


_default_ctx: dict = {}
_default_ctx_lock = threading.Lock()


def context_for(device: int) -> Context:
    """The process-wide context of ``device`` (one per device, made on first use; ``multi.DevicePool`` runs one
    host thread per device through these)."""
    dev = int(device)
    with _default_ctx_lock:
        ctx = _default_ctx.get(dev)
        if ctx is None:
            ctx = _default_ctx[dev] = Context(dev)
    return ctx


def default_context() -> Context:
    """Process-wide context on ``VND_DEVICE`` / ``LOCAL_RANK`` / device 0."""
    dev = int(os.environ.get('VND_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    n = device_count()
    if n > 0:
        dev %= n
    return context_for(dev)
