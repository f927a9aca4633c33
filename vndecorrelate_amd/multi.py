"""Several GPUs from ONE process: the batched many-stream mode behind the drop-in calls.

The reference user calls one function in one process (README.md:52-60); the streams of a batch are
independent (the channel loop couples nothing else, decorrelation.py:649), so
``convolve_velvet_noise_batched(x, fir, devices='all')`` and ``VelvetNoise.decorrelate_batched(x,
devices=...)`` cut the batch into contiguous blocks (``vnd_shard_range``: the remainder goes one each to
the first devices), give every device its block through that device's own ``vnd_ctx`` - one host thread
per device, each inside the ordinary pipelined ``*_host`` entry point (ctypes drops the GIL; the contexts
are distinct, so the calls overlap) - and write the blocks straight into one result array.  No collective
sits on the data path.  The tap table is built ONCE: uploaded to the first device and, with more than one
device, broadcast from there to the others over RCCL (a single-process communicator from
``ncclCommInitAll``, one ``vnd_taps_broadcast_rccl`` per device thread: 488 B at C = 2, K = 30, xGMI);
``table_transport='upload'`` deserialises the same image on every device instead.

``torch`` is not used here.  One process per GPU (``distributed.ShardedDecorrelator``, ``bench.py --gpus
N``) remains the form for jobs that keep their shards resident; this is the convenience form for hosts
that hold the whole batch in host memory.

A C / Go / JVM host does the same against ``include/vnd_amd.h``: ``vnd_device_count``, one ``vnd_ctx_create``
per device, ``vnd_shard_range`` per device index, one thread per device in ``vnd_convolve_f32_host`` on its
block (``INTEGRATION.md``).
"""
from __future__ import annotations

import ctypes
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np

from .distributed import shard_range
from .taps import TapArrays

Devices = Union[str, Sequence[int], None]


def resolve_devices(devices: Devices, available: Optional[int] = None) -> List[int]:
    """``'all'`` -> every visible device; a sequence -> those indices (validated, order kept, no repeats)."""
    if available is None:
        from . import _native
        available = _native.device_count()
    if isinstance(devices, str):
        if devices != 'all':
            raise ValueError(f"devices must be 'all' or a sequence of device indices, got {devices!r}")
        if available <= 0:
            raise RuntimeError('no HIP device visible: the velvet-noise kernels need an MI355X (gfx950)')
        return list(range(available))
    out = [int(d) for d in devices]
    if not out:
        raise ValueError('devices is empty')
    if len(set(out)) != len(out):
        raise ValueError(f'devices lists a device twice: {out}')
    for d in out:
        if not 0 <= d < available:
            raise ValueError(f'device {d} out of range (0..{available - 1})')
    return out


def blocks(total: int, parts: int) -> List[Tuple[int, int]]:
    """``(first, count)`` of every part's contiguous block (``distributed.shard_range`` = ``vnd_shard_range``)."""
    return [shard_range(total, parts, r) for r in range(parts)]


class _GpuWorker:
    """One device of the pool: its context and its copy of the table."""

    def __init__(self, ctx, table):
        self.ctx, self.table = ctx, table

    def convolve(self, x: np.ndarray, out: np.ndarray, mode: int) -> None:
        self.table.convolve_host(x, mode, out=out)

    def decorrelate(self, x: np.ndarray, out: np.ndarray, mode: int, **kw) -> None:
        self.table.decorrelate_host(x, mode, out=out, **kw)

    def close(self) -> None:
        self.table.close()


class _Rccl:
    """The three RCCL calls a single-process table broadcast needs, through ctypes (no torch)."""

    def __init__(self):
        lib = None
        for name in ('librccl.so.1', 'librccl.so'):
            try:
                lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                break
            except OSError:
                continue
        if lib is None:
            raise RuntimeError("librccl.so could not be loaded: pass table_transport='upload' to replicate the tap table without RCCL")
        lib.ncclCommInitAll.restype = ctypes.c_int
        lib.ncclCommInitAll.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        lib.ncclCommDestroy.restype = ctypes.c_int
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        self.lib = lib

    def init_all(self, devices: Sequence[int]) -> List[int]:
        comms = (ctypes.c_void_p * len(devices))()
        devs = (ctypes.c_int * len(devices))(*devices)
        rc = self.lib.ncclCommInitAll(comms, len(devices), devs)
        if rc != 0:
            raise RuntimeError(f'ncclCommInitAll over devices {list(devices)} failed: {self.lib.ncclGetErrorString(rc).decode()}')
        return [int(c) for c in comms]

    def destroy(self, comms: Sequence[int]) -> None:
        for c in comms:
            self.lib.ncclCommDestroy(ctypes.c_void_p(c))


class DevicePool:
    """The devices one process spreads a batch over.

    ``worker_factory(devices, arrays) -> [worker per device]`` builds the compute objects for one tap
    table (``worker.convolve(x_block, out_block, mode)``, ``worker.decorrelate(...)``); the default
    replicates the table on the GPUs.  Tests on a CPU-only box inject a checker here - there is no CPU
    path in the product."""

    def __init__(self, devices: Sequence[int], *, table_transport: str = 'rccl',
                 worker_factory: Optional[Callable] = None, cache_tables: int = 8):
        if table_transport not in ('rccl', 'upload'):
            raise ValueError("table_transport must be 'rccl' or 'upload'")
        self.devices = list(devices)
        self.table_transport = table_transport
        self._factory = worker_factory or self._gpu_workers
        self._threads = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix='vnd-dev')
        self._lock = threading.Lock()
        self._workers: 'dict[bytes, list]' = {}
        self._order: List[bytes] = []
        self._cache_tables = cache_tables
        self._rccl: Optional[_Rccl] = None
        self._comms: Optional[List[int]] = None
        self.last_blocks: List[Tuple[int, int]] = []
        self.last_transport: Optional[str] = None

    # ---- the table, once per device ------------------------------------------------
    def _gpu_workers(self, devices: Sequence[int], arrays: TapArrays) -> list:
        from . import _native
        ctxs = [_native.context_for(d) for d in devices]
        first = _native.TapTable.create(ctxs[0], arrays.tap_offsets, arrays.tap_index, arrays.tap_weight, **arrays.kwargs())
        if len(devices) == 1:
            self.last_transport = 'upload (one device)'
            return [_GpuWorker(ctxs[0], first)]
        if self.table_transport == 'upload':
            image = first.to_bytes()
            self.last_transport = 'upload'
            return [_GpuWorker(ctxs[0], first)] + [_GpuWorker(c, _native.TapTable.from_bytes(c, image)) for c in ctxs[1:]]
        # RCCL: one communicator per device in this process; every device's thread enters the broadcast (root = the first device).
        # The table image is 8 bytes per tap: if the communicator cannot be made (no librccl, a fabric the single-process
        # ncclCommInitAll refuses) or the broadcast fails, every device deserialises the image itself - the same bytes, still on
        # the GPUs, a warning and `last_transport` say which way it went.  A compute failure is never retried this way.
        def upload(why: str) -> list:
            import warnings
            warnings.warn(f'vndecorrelate_amd.multi: table broadcast over RCCL unavailable ({why}); uploading the table to each device instead')
            image = first.to_bytes()
            self.last_transport = f'upload (rccl unavailable: {why})'
            return [_GpuWorker(ctxs[0], first)] + [_GpuWorker(c, _native.TapTable.from_bytes(c, image)) for c in ctxs[1:]]
        try:
            if self._comms is None:
                self._rccl = self._rccl or _Rccl()
                self._comms = self._rccl.init_all(devices)
        except (OSError, RuntimeError, AttributeError) as exc:
            self._comms = None
            return upload(str(exc)[:160])
        comms = self._comms

        def receive(rank):
            return _native.TapTable.broadcast_rccl(ctxs[rank], first if rank == 0 else None, 0, rank, comms[rank])
        try:
            tables = list(self._threads.map(receive, range(len(devices))))
        except Exception as exc:
            return upload(f'broadcast failed: {str(exc)[:140]}')
        self.last_transport = 'rccl'
        return [_GpuWorker(c, t) for c, t in zip(ctxs, tables)]

    def workers(self, arrays: TapArrays) -> list:
        key = arrays.to_bytes()
        with self._lock:
            found = self._workers.get(key)
            if found is not None:
                self._order.remove(key)
                self._order.append(key)
                return found
            made = self._factory(self.devices, arrays)
            if len(made) != len(self.devices):
                raise RuntimeError('the worker factory must return one worker per device')
            self._workers[key] = made
            self._order.append(key)
            while len(self._order) > self._cache_tables:
                # dropped, never closed here: a thread still inside a call holds a reference (TapTable.__del__ frees the device copy)
                self._workers.pop(self._order.pop(0), None)
            return made

    # ---- the batch, a block per device ---------------------------------------------
    def map_streams(self, arrays: TapArrays, x: np.ndarray, out: np.ndarray, op: str, mode: int, **kw) -> np.ndarray:
        """``op`` ('convolve' | 'decorrelate') of every device's block of ``x`` ``(B, n, Cx)`` into the same block of
        ``out`` ``(B, n, C)``; blocks run side by side, one host thread per device with streams to do."""
        if x.ndim != 3 or out.ndim != 3 or out.shape[:2] != x.shape[:2]:
            raise ValueError(f'expected (batch, n, C) arrays of one batch, got {x.shape} and {out.shape}')
        workers = self.workers(arrays)
        cut = blocks(x.shape[0], len(self.devices))
        self.last_blocks = cut                            # (what the last call did: for tests and tools; callers may overlap)

        def run(part):
            first, count = cut[part]
            if count:
                getattr(workers[part], op)(x[first:first + count], out[first:first + count], mode, **kw)
        busy = [p for p, (_, count) in enumerate(cut) if count]
        if len(busy) == 1:
            run(busy[0])                                  # (no thread hop for one block)
        else:
            errors = []
            for fut in [self._threads.submit(run, p) for p in busy]:
                try:
                    fut.result()
                except Exception as exc:                  # every block finishes (or fails) before the call returns: `out` is the caller's
                    errors.append(exc)
            if errors:
                raise errors[0]
        return out

    def close(self) -> None:
        with self._lock:
            for made in self._workers.values():
                for w in made:
                    try:
                        w.close()
                    except Exception:
                        pass
            self._workers.clear()
            self._order.clear()
            if self._comms is not None and self._rccl is not None:
                self._rccl.destroy(self._comms)
            self._comms = None
        self._threads.shutdown(wait=True)


_pools: 'dict[tuple, DevicePool]' = {}
_pools_lock = threading.Lock()


def pool_for(devices: Devices, *, table_transport: Optional[str] = None) -> DevicePool:
    """The process-wide pool of a device list (contexts, table copies and, with RCCL, the communicator are kept)."""
    import os
    transport = table_transport or os.environ.get('VND_TABLE_TRANSPORT', 'rccl')
    ids = tuple(resolve_devices(devices))
    with _pools_lock:
        pool = _pools.get((ids, transport))
        if pool is None:
            pool = _pools[(ids, transport)] = DevicePool(ids, table_transport=transport)
    return pool


def close_pools() -> None:
    with _pools_lock:
        pools = list(_pools.values())
        _pools.clear()
    for p in pools:
        p.close()
