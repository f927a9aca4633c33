"""Batched many-stream mode across the GPUs of one node.

The hot path shards by *independent streams* (reference: the channel loop and
nothing else couples samples, decorrelation.py:649): every rank
owns a contiguous block of the batch, runs the same kernels on its own GPU, and
no collective sits on the data path.  The only communication is one broadcast of
the shared tap-table image (8*K*C bytes + header) from the rank that built it
(and, when ONE long stream is cut over the ranks in time, one forward-halo
send/recv per rank: ``ShardedDecorrelator.convolve_time_shard``) -
``torch.distributed`` with backend ``nccl`` is RCCL over xGMI on ROCm; ``gloo``
works too (that is what the CPU tests use).

One process per GPU (``torchrun --nproc-per-node N``); torch is used for process
groups and device memory only.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from .taps import TapArrays


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of ``total`` streams: ``(start, count)`` of
    ``rank``; the remainder goes one each to the lowest ranks."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f'bad rank {rank} of {world_size}')
    base, extra = divmod(total, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def broadcast_bytes(payload: Optional[bytes], src: int = 0, group=None, device=None) -> bytes:
    """Every rank returns ``payload`` of rank ``src`` (two broadcasts: length, body).
    Tensors live on ``device`` (a CUDA device for nccl/RCCL, CPU for gloo)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if payload is None:
            raise ValueError('payload missing on a single-rank run')
        return payload
    rank = dist.get_rank(group)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) \
            if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    size = torch.tensor([len(payload) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src=src, group=group)
    body = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    if rank == src:
        body.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(body, src=src, group=group)
    return body.cpu().numpy().tobytes()


class ShardedDecorrelator:
    """Shared impulse table, streams sharded over the ranks of ``group``.

    ``arrays`` (the table) is needed on ``src`` only; other ranks receive it.
    ``backend(arrays) -> callable(x_local, mode)`` builds the per-rank compute
    object; the default uploads the table to this rank's GPU.  (Tests on a
    CPU-only box inject a checker here - there is no CPU path in the product.)
    """

    def __init__(self, arrays: Optional[TapArrays] = None, *, src: int = 0, group=None, device=None,
                 backend: Optional[Callable] = None):
        import torch.distributed as dist
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        image = arrays.to_bytes() if (arrays is not None and self.rank == src) else None
        if self.world_size == 1 and image is None:
            raise ValueError('the tap table is required on the source rank')
        self.image = broadcast_bytes(image, src=src, group=group, device=device)
        self.arrays = TapArrays.from_bytes(self.image)
        self._convolve = (backend or self._gpu_backend)(self.arrays)

    @staticmethod
    def _gpu_backend(arrays: TapArrays):
        from . import _native
        table = _native.TapTable.from_bytes(_native.default_context(), arrays.to_bytes())

        def run(x_local: np.ndarray, mode: int) -> np.ndarray:
            return table.convolve_host(np.ascontiguousarray(x_local, dtype=np.float32), mode)

        run.table = table
        return run

    def shard(self, total_streams: int) -> Tuple[int, int]:
        return shard_range(total_streams, self.world_size, self.rank)

    def convolve_local(self, x_local: np.ndarray, mode: int = 2) -> np.ndarray:
        """This rank's ``(b_local, n, C)`` block -> same shape.  No communication."""
        if x_local.ndim != 3:
            raise ValueError(f'expected (streams, n, C), got {x_local.shape}')
        if x_local.shape[0] == 0:
            return np.zeros(x_local.shape, np.float32)
        return self._convolve(x_local, mode)

    def prepare(self, streams_local: int, frames: int, in_channels: Optional[int] = None, mode: int = 2) -> None:
        """Build now the per-table kernel this rank's shard launches will use (``vnd_prepare_launch``).  A rank's
        shard of a batch is a SMALL launch (the N = 8 shard of 1024 one-second streams: 6 M frames), and small
        launches never stall for a hipRTC build themselves - call this once after the table broadcast, before the
        passes: the 128-stream pass then takes 26 us instead of the generic kernel's 30 (tools/closed/shard_try.py)."""
        table = getattr(self._convolve, 'table', None)
        if table is not None and streams_local > 0 and frames > 0:
            table.prepare(streams_local, frames, in_channels or table.num_channels, mode)

    def convolve_local_device(self, x_local, y_local=None, mode: int = 2, stream: Optional[int] = None):
        """Device-resident form of :meth:`convolve_local`: ``x_local`` is this rank's ``(b_local, n, C)``
        float32 block already on ITS GPU (a torch tensor); the kernels are enqueued on ``stream`` (default:
        torch's current stream) and the result tensor is returned without a host round trip - shards stay
        resident between the table broadcast and whatever consumes the output."""
        import torch
        table = getattr(self._convolve, 'table', None)
        if table is None:
            raise RuntimeError('the device-resident path needs the GPU backend')
        if x_local.dim() != 3 or x_local.dtype != torch.float32 or not x_local.is_contiguous() or not x_local.is_cuda:
            raise ValueError('expected a contiguous float32 CUDA tensor (streams, n, C)')
        b_local, n, c = x_local.shape
        if y_local is None:
            y_local = torch.empty((b_local, n, table.num_channels), dtype=torch.float32, device=x_local.device)
        if b_local:
            with torch.cuda.device(x_local.device):
                s = torch.cuda.current_stream().cuda_stream if stream is None else stream
                table.convolve_device(x_local.data_ptr(), y_local.data_ptr(), b_local, n, c, mode, s)
        return y_local

    def convolve_time_shard(self, x_local: np.ndarray, mode: int = 2, device=None) -> np.ndarray:
        """ONE long stream cut over the ranks in time: rank r holds the contiguous frames
        ``x_local`` ``(n_r, C)`` that follow rank r-1's.  Output frame n reads input frames
        n .. n + max_index (``x[n + i]``, decorrelation.py:656-658), so every rank needs the first
        ``max_index`` frames that FOLLOW its slice: one forward-halo exchange (point-to-point
        send/recv, RCCL over xGMI with backend ``nccl``: 11.5 KB at cfg2) and no other traffic
        (SURVEY.md 8e).  Slices shorter than the halo are handled (the halo then comes from
        several ranks); the stream's true end keeps the reference's dropped terms.  Returns this
        rank's ``(n_r, C)`` outputs - in exact mode bit-identical to the unsharded call."""
        import torch
        import torch.distributed as dist
        if x_local.ndim != 2:
            raise ValueError(f'expected (frames, C), got {x_local.shape}')
        x_local = np.ascontiguousarray(x_local, dtype=np.float32)
        n_local, channels = x_local.shape
        if not dist.is_initialized() or self.world_size == 1:
            return self._convolve(x_local[None], mode)[0] if n_local else np.zeros((0, self.arrays.num_channels), np.float32)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) \
                if dist.get_backend(self.group) == 'nccl' else torch.device('cpu')
        counts = torch.zeros(self.world_size, dtype=torch.int64, device=device)
        counts[self.rank] = n_local
        dist.all_reduce(counts, group=self.group)                       # every rank learns every slice length
        counts = [int(v) for v in counts.cpu()]
        starts = np.concatenate([[0], np.cumsum(counts)])
        halo = int(self.arrays.tap_index.max()) if len(self.arrays.tap_index) else 0

        def wanted(r):                                                  # the frames rank r needs past its own
            return int(starts[r + 1]), int(min(starts[r + 1] + halo, starts[-1]))

        out_channels = self.arrays.num_channels
        # only the frames other ranks ask for travel: the prefix of at most `halo` frames of this slice
        mine = torch.from_numpy(x_local[:min(n_local, halo)]).to(device)
        ops, pieces = [], []
        lo, hi = wanted(self.rank)
        for s in range(self.rank + 1, self.world_size):                 # receive: prefixes of later slices
            a, b = max(lo, int(starts[s])), min(hi, int(starts[s + 1]))
            if b > a:
                buf = torch.empty((b - a, channels), dtype=torch.float32, device=device)
                pieces.append(buf)
                ops.append(dist.P2POp(dist.irecv, buf, s, group=self.group))
        for r in range(self.rank):                                      # send: what earlier ranks need of mine
            rlo, rhi = wanted(r)
            a, b = max(rlo, int(starts[self.rank])), min(rhi, int(starts[self.rank + 1]))
            if b > a:
                part = mine[a - int(starts[self.rank]):b - int(starts[self.rank])].contiguous()
                ops.append(dist.P2POp(dist.isend, part, r, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if n_local == 0:
            return np.zeros((0, out_channels), np.float32)
        window = np.concatenate([x_local] + [p.cpu().numpy() for p in pieces]) if pieces else x_local
        return self._convolve(window[None], mode)[0, :n_local]

    def convolve_global(self, x_all: np.ndarray, mode: int = 2) -> np.ndarray:
        """Convenience for small jobs: every rank passes the same full batch and gets
        back only ITS block; stack blocks in rank order to rebuild the batch."""
        start, count = self.shard(x_all.shape[0])
        return self.convolve_local(x_all[start:start + count], mode)


def all_gather_blocks(block: np.ndarray, total: int, group=None, device=None) -> np.ndarray:
    """Concatenate the ranks' contiguous blocks (as cut by :func:`shard_range`) of a 1-D float64
    vector of length ``total``: one all_gather of blocks padded to the largest block."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(block, np.float64)
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) \
            if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    widest = shard_range(total, world, 0)[1]
    mine = torch.zeros(max(widest, 1), dtype=torch.float64, device=device)
    mine[:len(block)] = torch.as_tensor(np.asarray(block, np.float64), device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return np.concatenate([parts[r][:shard_range(total, world, r)[1]].cpu().numpy() for r in range(world)])


def sharded_grid_scan(input_signal: np.ndarray, decorrelators, *, group=None, device=None,
                      scorer: Optional[Callable] = None, **objective) -> np.ndarray:
    """The optimiser's candidate scan (``optimization.grid_scan``) over several GPUs: the
    candidates are independent, so rank r scores a contiguous block of them on its own GPU
    (every rank holds the signal) and one all_gather of the score blocks - RCCL with backend
    ``nccl`` - gives every rank the full score vector.  ``scorer(signal, candidates, **objective)``
    defaults to the device scan; CPU tests inject a checker."""
    import torch.distributed as dist
    decorrelators = list(decorrelators)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, count = shard_range(len(decorrelators), world, rank)
    if scorer is None:
        from .optimization import grid_scan as scorer
    mine = scorer(input_signal, decorrelators[start:start + count], **objective) if count else np.zeros(0)
    return all_gather_blocks(np.asarray(mine, np.float64), len(decorrelators), group=group, device=device)
