"""Tap tables: the host-side description of a velvet-noise filter bank that the
HIP kernels consume (layout documented at ``vnd_taps_create`` in
``include/vnd_amd.h``).

Two builders, one per reference path, because the two paths associate the same
taps differently and parity is defined per path (SURVEY.md §8 a1 vs a6):

* :func:`function_path_arrays` - what ``convolve_velvet_noise`` sees: the
  nonzeros of a dense ``(L, C)`` FIR in ascending index, weights as stored
  (reference ``decorrelation.py:651-654``).  Duplicate positions have already
  collapsed in the dense FIR (last write wins, ``:623``).
* :func:`class_path_arrays` - what ``VelvetNoise.convolve`` walks: per channel
  and envelope segment the negative taps, then the positive ones, in generation
  order, duplicates kept; one gain per segment (``decorrelation.py:402-414``).

Pure NumPy, no device access: testable on a CPU-only box.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np


@dataclass
class TapArrays:
    tap_offsets: np.ndarray                  # int32 [C+1]
    tap_index: np.ndarray                    # int32 [total]
    tap_weight: np.ndarray                   # float32 [total]
    seg_offsets: Optional[np.ndarray] = None  # int32 [C+1]
    seg_end: Optional[np.ndarray] = None      # int32 [S]  exclusive, absolute tap positions
    seg_gain: Optional[np.ndarray] = None     # float32 [S]
    chan_flags: Optional[np.ndarray] = None   # uint8 [C], bit0 = copy through
    apply_gain: bool = False

    @property
    def num_channels(self) -> int:
        return len(self.tap_offsets) - 1

    def kwargs(self) -> dict:
        return dict(seg_offsets=self.seg_offsets, seg_end=self.seg_end, seg_gain=self.seg_gain,
                    chan_flags=self.chan_flags, apply_gain=self.apply_gain)

    # ---- transport image: byte-identical to vnd_taps_serialize (csrc/vnd_amd.hip) ----
    # int32 header[8] = {magic "VNDT", image version (VND_TAPS_IMAGE_VERSION), C, taps, segments, has_seg, has_flags,
    # apply_gain}, then tap_offsets[C+1], tap_index, tap_weight, (seg_offsets[C+1], seg_end,
    # seg_gain), (chan_flags padded to 4 bytes).  This is what travels over RCCL.
    _MAGIC = 0x564E4454
    _IMAGE_VERSION = 1

    def to_bytes(self) -> bytes:
        channels, total = self.num_channels, len(self.tap_index)
        has_seg = self.seg_offsets is not None
        has_flags = self.chan_flags is not None
        segs = len(self.seg_end) if has_seg else 0
        head = np.array([self._MAGIC, self._IMAGE_VERSION, channels, total, segs, int(has_seg), int(has_flags),
                         int(bool(self.apply_gain))], np.int32)
        parts = [head, self.tap_offsets.astype(np.int32), self.tap_index.astype(np.int32),
                 self.tap_weight.astype(np.float32)]
        if has_seg:
            parts += [self.seg_offsets.astype(np.int32), self.seg_end.astype(np.int32),
                      self.seg_gain.astype(np.float32)]
        image = b''.join(np.ascontiguousarray(p).tobytes() for p in parts)
        if has_flags:
            flags = np.zeros((channels + 3) // 4 * 4, np.uint8)
            flags[:channels] = self.chan_flags
            image += flags.tobytes()
        return image

    @classmethod
    def from_bytes(cls, image: bytes) -> 'TapArrays':
        head = np.frombuffer(image, np.int32, 8)
        if head[0] != cls._MAGIC or head[1] != cls._IMAGE_VERSION:
            raise ValueError('not a tap-table image of this format version')
        channels, total, segs, has_seg, has_flags, gain = (int(v) for v in head[2:8])
        pos = 32

        def take(dtype, count):
            nonlocal pos
            out = np.frombuffer(image, dtype, count, pos).copy()
            pos += out.nbytes
            return out

        arrays = cls(take(np.int32, channels + 1), take(np.int32, total), take(np.float32, total))
        if has_seg:
            arrays.seg_offsets = take(np.int32, channels + 1)
            arrays.seg_end = take(np.int32, segs)
            arrays.seg_gain = take(np.float32, segs)
        if has_flags:
            arrays.chan_flags = take(np.uint8, channels)
        arrays.apply_gain = bool(gain)
        if arrays.tap_offsets[-1] != total:
            raise ValueError('corrupt tap-table image')
        return arrays


def function_path_arrays(fir: np.ndarray, num_channels: Optional[int] = None) -> TapArrays:
    """CSR table of the nonzeros of ``fir[:, c]`` for the first ``num_channels`` columns."""
    fir = np.asarray(fir)
    if fir.ndim == 1:
        fir = fir[:, None]
    channels = fir.shape[1] if num_channels is None else num_channels
    offsets = np.zeros(channels + 1, np.int32)
    idx, w = [], []
    for c in range(channels):
        col = fir[:, c]
        nz = np.flatnonzero(col != 0.0)
        idx.append(nz.astype(np.int32))
        w.append(col[nz].astype(np.float32))
        offsets[c + 1] = offsets[c] + len(nz)
    return TapArrays(offsets, np.concatenate(idx) if idx else np.zeros(0, np.int32),
                     np.concatenate(w) if w else np.zeros(0, np.float32))


def class_path_arrays(channels: Sequence, envelope: Sequence[float], apply_gain: bool) -> TapArrays:
    """``channels[c]`` is ``None`` for an unfiltered (copied-through) channel, else a
    sequence of segments, each ``(negative_indexes, positive_indexes)``.
    ``envelope[s]`` is read for every generated segment when ``apply_gain`` is true
    (an envelope shorter than the segment list raises IndexError, as upstream)."""
    tap_offsets, seg_offsets = [0], [0]
    idx, w, seg_end, seg_gain, flags = [], [], [], [], []
    for segs in channels:
        if segs is None:
            flags.append(1)
            tap_offsets.append(tap_offsets[-1])
            seg_offsets.append(seg_offsets[-1])
            continue
        flags.append(0)
        count = tap_offsets[-1]
        for s, (neg, pos) in enumerate(segs):
            idx.extend(int(i) for i in neg)
            w.extend(-1.0 for _ in neg)
            idx.extend(int(i) for i in pos)
            w.extend(1.0 for _ in pos)
            count += len(neg) + len(pos)
            seg_end.append(count)
            seg_gain.append(float(envelope[s]) if apply_gain else 1.0)
        tap_offsets.append(count)
        seg_offsets.append(len(seg_end))
    return TapArrays(np.asarray(tap_offsets, np.int32), np.asarray(idx, np.int32),
                     np.asarray(w, np.float32), np.asarray(seg_offsets, np.int32),
                     np.asarray(seg_end, np.int32), np.asarray(seg_gain, np.float32),
                     np.asarray(flags, np.uint8), bool(apply_gain))


def class_path_bank_arrays(members: Sequence) -> TapArrays:
    """The class-path bank of ``concat_tap_arrays([class_path_arrays(*m) for m in members])`` built in
    one pass over plain Python lists (``members``: ``(channels, envelope, apply_gain)`` triples): the
    candidate scan concatenates hundreds of small tables, and per-table NumPy calls were most of its
    host time."""
    tap_offsets, seg_offsets = [0], [0]
    idx, w, seg_end, seg_gain, flags = [], [], [], [], []
    any_gain = False
    count = 0
    for channels, envelope, apply_gain in members:
        any_gain = any_gain or bool(apply_gain)
        for segs in channels:
            if segs is None:
                flags.append(1)
                tap_offsets.append(count)
                seg_offsets.append(len(seg_end))
                continue
            flags.append(0)
            for s, (neg, pos) in enumerate(segs):
                idx += neg
                idx += pos
                w += [-1.0] * len(neg)
                w += [1.0] * len(pos)
                count += len(neg) + len(pos)
                seg_end.append(count)
                seg_gain.append(float(envelope[s]) if apply_gain else 1.0)
            tap_offsets.append(count)
            seg_offsets.append(len(seg_end))
    out = TapArrays(np.asarray(tap_offsets, np.int32), np.asarray(idx, np.int32), np.asarray(w, np.float32),
                    np.asarray(seg_offsets, np.int32), np.asarray(seg_end, np.int32), np.asarray(seg_gain, np.float32),
                    np.asarray(flags, np.uint8) if any(flags) else None, any_gain)
    return out


def concat_tap_arrays(tables: Sequence[TapArrays]) -> TapArrays:
    """Channel-wise concatenation: a bank of F tables of C channels each becomes one table
    of F*C channels (table f owns channels ``f*C .. f*C+C-1``), the shape the fan-out
    launch wants (``vnd_convolve_fanout_*``: output channel c reads input channel c % C).
    All tables must be of the same kind (function path, or class path with segments).
    A class-path bank applies gains as soon as one member does; the others carry gain 1.0,
    and multiplying by 1.0 is exact."""
    tables = list(tables)
    if not tables:
        raise ValueError('empty filter bank')
    with_seg = [t.seg_offsets is not None for t in tables]
    if any(with_seg) and not all(with_seg):
        raise ValueError('cannot mix function-path and class-path tables in one bank')
    tap_offsets, idx, w = [np.zeros(1, np.int32)], [], []
    seg_offsets, seg_end, seg_gain, flags = [np.zeros(1, np.int32)], [], [], []
    tap_base = seg_base = 0
    for t in tables:
        tap_offsets.append(t.tap_offsets[1:].astype(np.int64) + tap_base)
        idx.append(t.tap_index)
        w.append(t.tap_weight)
        if with_seg[0]:
            seg_offsets.append(t.seg_offsets[1:].astype(np.int64) + seg_base)
            seg_end.append(t.seg_end.astype(np.int64) + tap_base)
            seg_gain.append(t.seg_gain)
            seg_base += len(t.seg_end)
        flags.append(t.chan_flags if t.chan_flags is not None else np.zeros(t.num_channels, np.uint8))
        tap_base += len(t.tap_index)
    out = TapArrays(np.concatenate(tap_offsets).astype(np.int32), np.concatenate(idx).astype(np.int32),
                    np.concatenate(w).astype(np.float32))
    if with_seg[0]:
        out.seg_offsets = np.concatenate(seg_offsets).astype(np.int32)
        out.seg_end = np.concatenate(seg_end).astype(np.int32)
        out.seg_gain = np.concatenate(seg_gain).astype(np.float32)
        out.apply_gain = any(t.apply_gain for t in tables)
    if any(t.chan_flags is not None for t in tables):
        out.chan_flags = np.concatenate(flags).astype(np.uint8)
    return out
