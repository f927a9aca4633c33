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
