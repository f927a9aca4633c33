"""Drop-in host layer for the velvet-noise path of ckonst/VNDecorrelate.

Same public names, keyword arguments, defaults and exceptions as the
reference's ``vndecorrelate.decorrelation`` (v1.1.0; ``file:line`` citations
are relative to its checkout), but every tap sum runs on an MI355X through the
C ABI in ``include/vnd_amd.h``:

===========================  ==================================================
reference                    here
===========================  ==================================================
``convolve_velvet_noise``    :func:`convolve_velvet_noise`  -> ``vnd_convolve_f32_host``
``generate_velvet_noise``    :func:`generate_velvet_noise`  (host NumPy, O(K))
``VelvetNoise``              :class:`VelvetNoise` (``convolve`` on the GPU)
``SignalChain``              :class:`SignalChain`
``HaasEffect``               :class:`HaasEffect` (device kernel ``vnd_haas_f64_*`` in a
                             device-resident chain, NumPy otherwise)
``WhiteNoise``               NumPy-only chain stage (out of the GPU scope, SURVEY.md §2
                             row 7; kept so chains stay drop-in)
===========================  ==================================================

There is no CPU implementation of the tap sum in this package: without the
built extension or without a gfx950 device the calls raise ``RuntimeError``.

Operand types (DESIGN.md "Parity"): in the exact mode every dtype combination of
the function path is bit-identical to the reference - float32 and the integer
types NumPy promotes to float32 through the float32 kernels, float64 / int32 /
int64 signals and float64 filters through ``vnd_convolve_promote_host``, which
rounds to float32 at every tap as ``out += x * value`` does.  In the tolerance
modes those operands are rounded to float32 first.

Threading: the functions are re-entrant like the reference's; calls that share a
context are serialised inside the library (one stream and one set of staging
buffers per context).
"""
from __future__ import annotations

import hashlib
import threading
from abc import ABC, abstractmethod
from collections import OrderedDict
from dataclasses import dataclass, field
from functools import partial
from typing import Any, Callable, Iterator, List, Optional, Protocol, Sequence

import numpy as np
from numpy.typing import NDArray

from . import _native
from .taps import TapArrays, class_path_arrays, concat_tap_arrays, function_path_arrays
from .utils.dsp import (
    IDENTITY_ENVELOPE,
    LayoutMode,
    LR_to_MS,
    MS_to_LR,
    apply_log_distribution,
    apply_stereo_width,
    check_equal_length,
    encode_signal_to_side_channel,
    generate_log_distribution,
    mono_to_stereo,
    rms_normalize,
    to_float32,
)

DEFAULT_SEGMENT_ENVELOPE = (0.85, 0.55, 0.35, 0.2)

# Arithmetic used by the host API.  EXACT reproduces the reference bit for bit
# for float32 inputs; FMA is the single-rounding variant (<= 1e-6 of peak).
MODE_EXACT = _native.MODE_EXACT
MODE_FMA = _native.MODE_FMA
MODE_FAST = _native.MODE_FAST
_default_mode = MODE_EXACT


_device_epilogue: Optional[bool] = None


def set_device_epilogue(enabled: Optional[bool]) -> None:
    """Where ``VelvetNoise.decorrelate``'s epilogue (side-channel encode, width, RMS
    normalise) runs.

    ``None`` (default): on the GPU behind the convolution, with the normaliser's sums of
    squares in NumPy's own order (a sequential float32 recurrence for ``(n, C >= 2)`` arrays,
    reproduced bit for bit).  In MODE_EXACT the whole stage is then bit-identical to the
    reference; in the other modes it differs from it only through the convolution
    (~1e-6 of peak).  A single-channel table's sums are NumPy's pairwise ones (8192-sample
    chunks, 128-sample leaves), reproduced the same way; custom normalisers keep the host epilogue.
    ``True``: on the GPU in its fastest form - in MODE_FAST everything fused into the fast
    kernel with exactly rounded float64 sums, which puts the normalised output ~1e-4
    relative from the reference's on long signals (NumPy's sequential sum is that far off).
    ``False``: in NumPy on the host, behind the device convolution."""
    global _device_epilogue
    _device_epilogue = None if enabled is None else bool(enabled)


def _use_device_epilogue(num_outs: int, has_normalizer: bool, frames: int = 0, c_contiguous: bool = True) -> bool:
    """The default policy: the device epilogue wherever it repeats what NumPy would do."""
    if _device_epilogue is not None:
        return _device_epilogue
    if not has_normalizer:
        return True                                       # the pointwise steps are always NumPy's
    # NumPy's row-by-row float32 sums are repeated on the device for 2 to 32 channels of a C-contiguous
    # signal, and so are the pairwise sums it forms for a single channel (other memory layouts: host)
    return 1 <= num_outs <= 32 and c_contiguous


def _normalize_flag(has_normalizer: bool) -> int:
    if not has_normalizer:
        return _native.NORMALIZE_OFF
    return _native.NORMALIZE_RMS if _device_epilogue else _native.NORMALIZE_RMS_REFERENCE_ORDER


def set_default_mode(mode: int) -> None:
    """Choose the arithmetic of subsequent host-API calls (MODE_EXACT / MODE_FMA)."""
    global _default_mode
    if mode not in (MODE_EXACT, MODE_FMA, MODE_FAST):
        raise ValueError(f'unknown mode {mode}')
    _default_mode = mode


# ----------------------------------------------------------------------------
# Protocols / abstract bases   (decorrelation.py:31-59)
# ----------------------------------------------------------------------------
class SignalProcessor(Protocol):
    sample_rate_hz: int
    num_outs: int

    def __call__(self, input_signal: NDArray) -> NDArray: ...


class StatelessDecorrelator(Protocol):
    def __call__(self, input_signal: NDArray, **kwargs) -> NDArray: ...


@dataclass(kw_only=True)
class Decorrelator(ABC):
    """Base of the stateful stages: ``stage(x)`` is ``stage.decorrelate(x)``."""

    sample_rate_hz: int
    num_outs: int = 2
    width: Optional[float] = None

    @abstractmethod
    def decorrelate(self, input_signal: NDArray) -> NDArray:
        raise NotImplementedError

    def __call__(self, input_signal: NDArray) -> NDArray:
        return self.decorrelate(input_signal)


# ----------------------------------------------------------------------------
# tap placement shared by both generators   (decorrelation.py:488-523, :573-611)
# ----------------------------------------------------------------------------
def _draw_taps(seed, num_impulses: int, num_filters: int, fir_length: int, sample_rate_hz,
               duration_seconds: float, strength: float):
    """Positions ``int32 (K+1, F)`` and signs ``(K, F)`` in {-1, +1}.

    The order of the two uniform draws (signs, then offsets) and their shapes
    are part of the contract: they fix which PCG64 values land where.
    """
    rng = np.random.default_rng(seed)
    weights = generate_log_distribution(strength, num_impulses)
    marks = np.cumsum(weights)
    if strength == 0.0:
        marks -= 1.0                      # uniform case starts at sample 0
    marks *= fir_length / marks[-1]
    sign_draw = rng.uniform(low=0, high=1, size=(num_impulses, num_filters))
    offset_draw = rng.uniform(low=0, high=1, size=(num_impulses + 1, num_filters))
    signs = (2 * np.round(sign_draw)) - 1
    mean_gap = sample_rate_hz / (num_impulses / duration_seconds)
    positions = np.stack([apply_log_distribution(offset_draw[:, f], weights, marks, mean_gap)
                          for f in range(num_filters)], axis=1)
    return positions, signs


def _segment_index(k: int, num_impulses: int, num_segments: int) -> int:
    return int(k / (num_impulses / num_segments))


def generate_velvet_noise(*, duration_seconds: float, num_impulses: int, num_outs: int = 2,
                          sample_rate_hz: int = 44100,
                          segment_envelope: Sequence[float] = DEFAULT_SEGMENT_ENVELOPE,
                          log_distribution_strength: float = 1.0,
                          seed: Optional[int] = None) -> NDArray:
    """Dense seeded velvet-noise FIR, float32 ``(int(duration*fs), num_outs)``.

    Mirrors ``generate_velvet_noise`` (decorrelation.py:549-627): the length
    truncates, a later impulse on an occupied sample overwrites the earlier one.
    Feed the result to :func:`convolve_velvet_noise`.
    """
    fir_length = int(duration_seconds * sample_rate_hz)
    envelope = tuple(segment_envelope) if len(segment_envelope) else IDENTITY_ENVELOPE
    fir = np.zeros((fir_length, num_outs), dtype=np.float32)
    positions, signs = _draw_taps(seed, num_impulses, num_outs, fir_length, sample_rate_hz,
                                  duration_seconds, log_distribution_strength)
    for c in range(num_outs):
        for k in range(num_impulses):
            fir[positions[k, c], c] = signs[k, c] * envelope[_segment_index(k, num_impulses, len(envelope))]
    return fir


# ----------------------------------------------------------------------------
# device tap-table cache for the stateless path
# ----------------------------------------------------------------------------
class _TableCache:
    """Small LRU of device tables keyed by FIR content, so calling the stateless
    function repeatedly with one FIR uploads its 8*K*C bytes once."""

    def __init__(self, capacity: int = 16):
        self.capacity = capacity
        self._items: 'OrderedDict[tuple, _native.TapTable]' = OrderedDict()
        self._lock = threading.Lock()

    def get(self, key, build: Callable[[], TapArrays]) -> _native.TapTable:
        with self._lock:
            table = self._items.get(key)
            if table is not None:
                self._items.move_to_end(key)
                return table
        arrays = build()
        table = _native.TapTable.create(_native.default_context(), arrays.tap_offsets,
                                        arrays.tap_index, arrays.tap_weight, **arrays.kwargs())
        with self._lock:
            table = self._items.setdefault(key, table)      # another thread may have built it meanwhile
            self._items.move_to_end(key)
            # An evicted table is only dropped, never closed here: a thread still inside a call
            # holds a reference, and the device memory goes when the last one does (TapTable.__del__).
            while len(self._items) > self.capacity:
                self._items.popitem(last=False)
        return table

    def clear(self):
        with self._lock:
            self._items.clear()


_fir_tables = _TableCache()


try:                                   # a 128-bit content hash of the filter per call: xxh3 takes 0.7 us for a 30 ms stereo
    from xxhash import xxh3_128_digest as _digest16          # filter where blake2b takes 16 (of a 190 us call)
except ImportError:                    # pragma: no cover - the hash is an optional dependency
    def _digest16(view):
        return hashlib.blake2b(view.tobytes(), digest_size=16).digest()


def _fir_key(fir: np.ndarray, channels: int):
    view = np.ascontiguousarray(fir[:, :channels])
    return (view.shape, str(view.dtype), _digest16(view), _native.default_context().device)


class _ArraysCache:
    """Function-path tap arrays (and their transport image) of the filters last spread over a device list: the several-device
    call otherwise rebuilds them - a scan of the dense FIR and a serialisation - on every call.  Keyed by the filter's content."""

    def __init__(self, capacity: int = 32):
        self.capacity, self._lock, self._items = capacity, threading.Lock(), OrderedDict()

    def get(self, fir: np.ndarray, channels: int):
        view = np.ascontiguousarray(fir[:, :channels])
        key = (view.shape, str(view.dtype), _digest16(view))
        with self._lock:
            found = self._items.get(key)
            if found is not None:
                self._items.move_to_end(key)
                return found
        arrays = function_path_arrays(fir, channels)
        image = arrays.to_bytes()
        arrays.to_bytes = lambda: image                      # (immutable from here on: the pool keys its device copies by the image)
        with self._lock:
            self._items[key] = arrays
            while len(self._items) > self.capacity:
                self._items.popitem(last=False)
        return arrays


_fir_arrays = _ArraysCache()


def _promoted_convolve(x: NDArray, fir: NDArray, num_channels: int, mode: int) -> Optional[NDArray]:
    """The operand types NumPy multiplies in float64 (a float64 or int32/int64 signal, or any
    signal with a float64 filter such as ``VelvetNoise.FIR``): in the exact mode they take the
    promoting kernel, which rounds to float32 at every tap exactly as ``out += x * value`` does
    (decorrelation.py:656-658).  None when the float32 kernels apply."""
    if mode != MODE_EXACT or np.result_type(x.dtype, fir.dtype) != np.float64 or x.size == 0:
        return None
    offsets = np.zeros(num_channels + 1, np.int32)
    idx, weights = [], []
    for c in range(num_channels):
        nz = np.flatnonzero(fir[:, c] != 0.0)
        idx.append(nz.astype(np.int32))
        weights.append(fir[nz, c].astype(np.float64))
        offsets[c + 1] = offsets[c] + len(nz)
    xin = x if x.dtype in (np.float32, np.float64) else x.astype(np.float64)
    return _native.convolve_promote_host(_native.default_context(), np.ascontiguousarray(xin), offsets,
                                         np.concatenate(idx), np.concatenate(weights))


def convolve_velvet_noise(input_signal: NDArray, velvet_noise_filters: NDArray, *,
                          mode: Optional[int] = None) -> NDArray:
    """Stateless sparse convolution ``y[n,c] = sum_k w[c,k] * x[n + i[c,k], c]``.

    Drop-in for ``convolve_velvet_noise`` (decorrelation.py:630-660): same
    shapes in and out (float32 ``(n, C)`` result), ``ValueError`` when a
    multi-channel signal and the filters disagree on channel count, and the
    reference's ``IndexError`` for a 1-D signal.  ``mode`` picks the arithmetic
    (default: the bit-exact one).  Runs on the GPU; raises ``RuntimeError``
    without one.
    """
    fir = np.asarray(velvet_noise_filters)
    if fir.ndim == 1:
        fir = fir[:, None]
    if input_signal.ndim == 1:
        # The reference indexes the 1-D signal with two subscripts at its first tap.
        if np.any(fir[:, 0] != 0.0):
            raise IndexError('too many indices for array: array is 1-dimensional, but 2 were indexed')
        return np.zeros(input_signal.shape, dtype=np.float32)
    num_channels = input_signal.shape[1]
    if num_channels > 1:
        check_equal_length(input_signal, fir, dim=1)
    mode = _default_mode if mode is None else mode
    if num_channels:
        promoted = _promoted_convolve(np.asarray(input_signal), fir, num_channels, mode)
        if promoted is not None:
            return promoted
    x = np.ascontiguousarray(input_signal, dtype=np.float32)
    if x.shape[0] == 0 or num_channels == 0:
        return np.zeros(x.shape, dtype=np.float32)
    table = _fir_tables.get(_fir_key(fir, num_channels),
                            lambda: function_path_arrays(fir, num_channels))
    return table.convolve_host(x, mode)


def convolve_velvet_noise_batched(input_signals: NDArray, velvet_noise_filters: NDArray, *,
                                  mode: Optional[int] = None, devices=None) -> NDArray:
    """Many independent streams with one shared filter bank: ``(B, n, C)`` in and
    out, one kernel launch.  Equals stacking :func:`convolve_velvet_noise` over B.

    ``devices``: ``None`` - the process's default device; ``'all'`` or a list of device indices - the
    batch is cut into contiguous blocks of streams, one per device, run side by side from this one
    process (``multi.DevicePool``: the table is built once and replicated - over RCCL with more than one
    device -, no collective on the data path); the result is the same array either way."""
    if input_signals.ndim != 3:
        raise ValueError(f'expected (batch, n, C), got shape {input_signals.shape}')
    fir = np.asarray(velvet_noise_filters)
    if fir.ndim == 1:
        fir = fir[:, None]
    num_channels = input_signals.shape[2]
    if num_channels > 1 and fir.shape[1] != num_channels:
        raise ValueError('Input length mismatch: Expected signals of equal length, but got lengths '
                         f'{num_channels} and {fir.shape[1]} for dimension 1.')
    mode = _default_mode if mode is None else mode
    if num_channels:
        promoted = _promoted_convolve(np.asarray(input_signals), fir, num_channels, mode)
        if promoted is not None:
            return promoted
    x = np.ascontiguousarray(input_signals, dtype=np.float32)
    if x.size == 0:
        return np.zeros(x.shape, dtype=np.float32)
    if devices is not None:
        from . import multi
        arrays = _fir_arrays.get(fir, num_channels)
        out = _native.pinned_pool.empty(x.shape[:-1] + (arrays.num_channels,), np.float32)
        return multi.pool_for(devices).map_streams(arrays, x, out, 'convolve', mode)
    table = _fir_tables.get(_fir_key(fir, num_channels),
                            lambda: function_path_arrays(fir, num_channels))
    return table.convolve_host(x, mode)


def convolve_velvet_noise_bank(input_signal: NDArray, filter_bank: Sequence[NDArray], *,
                               mode: Optional[int] = None) -> NDArray:
    """One ``(n, C)`` signal through F filters ``(L_f, C)`` in a single launch; returns
    ``(F, n, C)`` float32 (a transposed view of the device result) with
    ``out[f] == convolve_velvet_noise(input_signal, filter_bank[f])``, bit for bit in the
    exact mode.  This is the candidate scan of the reference's optimiser
    (optimization.py:107-117 runs the F convolutions one after the other): the signal is
    uploaded once and every tile is staged per filter from L2, not from the host."""
    if input_signal.ndim != 2:
        raise ValueError(f'expected a (n, C) signal, got shape {input_signal.shape}')
    firs = [np.asarray(f) if np.asarray(f).ndim == 2 else np.asarray(f)[:, None] for f in filter_bank]
    if not firs:
        raise ValueError('empty filter bank')
    num_channels = input_signal.shape[1]
    for fir in firs:
        if num_channels > 1:
            check_equal_length(input_signal, fir, dim=1)
    x = np.ascontiguousarray(input_signal, dtype=np.float32)
    if x.shape[0] == 0 or num_channels == 0:
        return np.zeros((len(firs),) + x.shape, dtype=np.float32)
    key = ('bank', num_channels) + tuple(_fir_key(fir, num_channels) for fir in firs)
    table = _fir_tables.get(key, lambda: concat_tap_arrays([function_path_arrays(fir, num_channels)
                                                             for fir in firs]))
    y = table.convolve_host(x, _default_mode if mode is None else mode)       # (n, F*C)
    return y.reshape(x.shape[0], len(firs), num_channels).transpose(1, 0, 2)


# ----------------------------------------------------------------------------
# Velvet-noise impulse containers   (decorrelation.py:240-323)
# ----------------------------------------------------------------------------
@dataclass
class VelvetNoiseSegment:
    """Impulse positions of one envelope segment, split by sign."""

    negative_impulse_indexes: List[int] = field(default_factory=list)
    positive_impulse_indexes: List[int] = field(default_factory=list)

    def __iter__(self) -> Iterator:
        yield self.negative_impulse_indexes, '__isub__'
        yield self.positive_impulse_indexes, '__iadd__'

    def __getitem__(self, key: int) -> List[int]:
        if key == 0:
            return self.negative_impulse_indexes
        if key == 1:
            return self.positive_impulse_indexes
        raise ValueError('Invalid key')

    def __setitem__(self, key: int, value) -> None:
        if key == 0:
            self.negative_impulse_indexes = value
        elif key == 1:
            self.positive_impulse_indexes = value
        else:
            raise ValueError('Invalid key')


@dataclass
class VelvetNoiseSequence:
    """The segments of one output channel."""

    segments: List[VelvetNoiseSegment] = field(default_factory=list)

    @classmethod
    def create(cls, *, num_segments: int) -> 'VelvetNoiseSequence':
        return cls(segments=[VelvetNoiseSegment() for _ in range(num_segments)])

    def __iter__(self):
        return iter(self.segments)

    def __len__(self):
        return len(self.segments)

    def __getitem__(self, key: int) -> VelvetNoiseSegment:
        return self.segments[key]

    def __setitem__(self, key: int, value) -> None:
        self.segments[key] = value


@dataclass
class ParallelVelvetNoise:
    """One sequence per output channel; an unfiltered channel is an empty list."""

    fir_length_samples: int
    output_channels: list = field(default_factory=list)

    @property
    def num_outs(self) -> int:
        return len(self.output_channels)

    @property
    def num_impluses(self) -> int:      # (sic) - the reference's spelling, counted on channel 0
        return sum(len(seg.negative_impulse_indexes) + len(seg.positive_impulse_indexes)
                   for channel in self.output_channels[0:1] for seg in channel)

    num_impulses = num_impluses

    def __iter__(self):
        return iter(self.output_channels)

    def __getitem__(self, key: int):
        return self.output_channels[key]

    def __setitem__(self, key: int, value) -> None:
        self.output_channels[key] = value


_VelvetNoiseSegment = VelvetNoiseSegment
_VelvetNoiseSequence = VelvetNoiseSequence
_ParallelVelvetNoise = ParallelVelvetNoise


# ----------------------------------------------------------------------------
# VelvetNoise   (decorrelation.py:326-546)
# ----------------------------------------------------------------------------
@dataclass(kw_only=True)
class VelvetNoise(Decorrelator):
    """Velvet-noise decorrelator with the reference's fields and defaults.

    The impulse table is drawn once at construction and again only when
    ``num_outs``, ``num_impulses`` or ``fir_length_samples`` change
    (decorrelation.py:368-379); ``segment_envelope`` is read at convolve time.
    ``convolve`` uploads the table to the GPU once and reuses it.
    """

    duration_seconds: float = 0.03
    num_impulses: int = 30
    segment_envelope: Sequence[float] = DEFAULT_SEGMENT_ENVELOPE
    log_distribution_strength: float = 1.0
    normalizer: Optional[Callable[[NDArray, NDArray], None]] = rms_normalize
    filtered_channels: Sequence[int] = (0, 1)
    mode: LayoutMode = LayoutMode.MS
    seed: Optional[int] = None

    _velvet_noise: Any = field(default=None, repr=False, compare=False)
    _device: Any = field(default=None, repr=False, compare=False)   # (impulse table, envelope key, device, TapTable)

    def __post_init__(self) -> None:
        if self.num_impulses >= self.fir_length_samples * 0.2:
            raise ValueError(
                f'Velvet Noise Filter of length {self.fir_length_samples} with {self.num_impulses} '
                f'impulses is not sparse. (density={self.density:.2f})\n'
                '\tnum_impulses must be less than 20% the FIR length in samples.')
        if not self.segment_envelope:
            self.segment_envelope = IDENTITY_ENVELOPE
        self._velvet_noise = self._generate()

    # ---- derived quantities --------------------------------------------------
    @property
    def density(self) -> float:
        """Impulses per second."""
        return self.num_impulses / self.duration_seconds

    @property
    def fir_length_samples(self) -> int:
        return int(round(self.sample_rate_hz * self.duration_seconds))

    @property
    def unfiltered_channels(self):
        return filter(lambda c: c not in self.filtered_channels, range(self.num_outs))

    @property
    def velvet_noise(self) -> ParallelVelvetNoise:
        vn = self._velvet_noise
        if (self.num_outs != vn.num_outs or self.num_impulses != vn.num_impluses
                or self.fir_length_samples != vn.fir_length_samples):
            self._velvet_noise = self._generate()
        return self._velvet_noise

    @property
    def FIR(self) -> NDArray:
        """Dense float64 ``(fir_length_samples, len(filtered_channels))`` view of
        the impulse table; a later impulse on an occupied sample wins."""
        filtered = [seq for seq in self.velvet_noise if len(seq)]
        fir = np.zeros((self.fir_length_samples, len(self.filtered_channels)))
        for f, sequence in enumerate(filtered):
            for s, segment in enumerate(sequence):
                for i in segment.negative_impulse_indexes:
                    fir[i, f] = self.segment_envelope[s] * -1
                for i in segment.positive_impulse_indexes:
                    fir[i, f] = self.segment_envelope[s] * 1
        return fir

    # ---- generation ----------------------------------------------------------
    def _generate(self) -> ParallelVelvetNoise:
        num_segments = len(self.segment_envelope)
        table = ParallelVelvetNoise(fir_length_samples=self.fir_length_samples)
        positions, signs = _draw_taps(self.seed, self.num_impulses, len(self.filtered_channels),
                                      self.fir_length_samples, self.sample_rate_hz,
                                      self.duration_seconds, self.log_distribution_strength)
        for channel in range(self.num_outs):
            if channel not in self.filtered_channels:
                table.output_channels.append([])
                continue
            # random columns are addressed by OUTPUT channel number, as upstream (:531)
            column, column_signs = positions[:, channel], signs[:, channel]
            sequence = VelvetNoiseSequence.create(num_segments=num_segments)
            for k in range(self.num_impulses):
                segment = sequence[_segment_index(k, self.num_impulses, num_segments)]
                segment[int((column_signs[k] + 1) / 2)].append(column[k])
            table.output_channels.append(sequence)
        return table

    # ---- the hot path --------------------------------------------------------
    def _tap_member(self):
        """``(channels, envelope, apply_gain)`` as ``taps.class_path_arrays`` / ``class_path_bank_arrays`` take them."""
        apply_gain = self.segment_envelope != IDENTITY_ENVELOPE
        channels = []
        for sequence in self.velvet_noise:
            if not len(sequence):
                channels.append(None)
            else:
                channels.append([(seg.negative_impulse_indexes, seg.positive_impulse_indexes)
                                 for seg in sequence])
        return channels, self.segment_envelope, apply_gain

    def _tap_arrays(self) -> TapArrays:
        return class_path_arrays(*self._tap_member())

    def _device_table(self) -> _native.TapTable:
        """Device image of the current impulse table + envelope, uploaded once.  The
        cache holds the impulse-table OBJECT (compared with ``is``), not its id: a
        regenerated table may be allocated at a freed table's address."""
        vn = self.velvet_noise
        env = self.segment_envelope
        env_key = (type(env).__name__, tuple(env))
        device = _native.default_context().device
        cached = self._device
        if cached is None or cached[0] is not vn or cached[1] != env_key or cached[2] != device:
            arrays = self._tap_arrays()
            if cached is not None:
                cached[3].close()
            table = _native.TapTable.create(_native.default_context(), arrays.tap_offsets, arrays.tap_index,
                                            arrays.tap_weight, **arrays.kwargs())
            self._device = cached = (vn, env_key, device, table)
        return cached[3]

    def convolve(self, input_signal: NDArray) -> NDArray:
        """Velvet-noise filter every ``filtered_channel`` of a ``(n, >= num_outs)``
        signal on the GPU; other output channels are copied through.  float32
        ``(n, num_outs)``, bit-identical to ``VelvetNoise.convolve``
        (decorrelation.py:393-415) for float32 input."""
        if input_signal.ndim != 2:
            raise IndexError('too many indices for array: convolve expects a (n, channels) signal')
        x = np.ascontiguousarray(input_signal[:, :self.num_outs], dtype=np.float32)
        if x.shape[1] != self.num_outs:
            raise IndexError(f'index {self.num_outs - 1} is out of bounds for axis 1 '
                             f'with size {input_signal.shape[1]}')
        table = self._device_table()
        if x.shape[0] == 0:
            return np.zeros((0, self.num_outs), dtype=np.float32)
        return table.convolve_host(x, _default_mode)

    def decorrelate(self, input_signal: NDArray) -> NDArray:
        """Full stage (decorrelation.py:417-442): float32 cast, mono->stereo, GPU convolution,
        then the epilogue - side-channel encode (MS mode), width, normaliser - on the device
        where that is faithful to the reference (see ``set_device_epilogue``), else in NumPy."""
        input_signal = to_float32(input_signal)
        if input_signal.ndim == 1 and self.num_outs == 2 and input_signal.shape[0] > 0:
            # a mono signal (mono_to_stereo, decorrelation.py:428-431): the device reads the one channel for both outputs (fan-out);
            # the (n, 2) copy the reference makes is materialised only where the host epilogue needs it - it costs more than the
            # whole device stage of a 10 s signal
            mono = np.ascontiguousarray(input_signal, dtype=np.float32)[:, None]
            if _use_device_epilogue(2, self.normalizer is not None, mono.shape[0], True) and \
                    (self.normalizer is None or self.normalizer is rms_normalize):
                return self._decorrelate_on_device(mono)
            output_signal = self._device_table().convolve_host(mono, _default_mode)
            return self._host_epilogue(mono_to_stereo(input_signal), output_signal)
        if input_signal.ndim == 1:
            input_signal = mono_to_stereo(input_signal)
        # NumPy's sum order follows the memory layout (a Fortran-ordered signal is summed pairwise,
        # column by column): the device repeats the C-contiguous order only, so by default other
        # layouts keep the host epilogue, which sees the caller's array as the reference does
        if _use_device_epilogue(self.num_outs, self.normalizer is not None, input_signal.shape[0],
                                input_signal.flags.c_contiguous) and self._device_epilogue_applies(input_signal):
            return self._decorrelate_on_device(input_signal)
        output_signal = self.convolve(input_signal)
        return self._host_epilogue(input_signal, output_signal)

    def _host_epilogue(self, input_signal: NDArray, output_signal: NDArray) -> NDArray:
        """decorrelation.py:433-440, in place on ``output_signal`` (NumPy: bit-identical)."""
        if self.mode == LayoutMode.MS:
            encode_signal_to_side_channel(input_signal, output_signal)
        if self.width is not None:
            apply_stereo_width(output_signal, self.width)
        if self.normalizer:
            self.normalizer(input_signal, output_signal)
        return output_signal


    # ---- device epilogue ------------------------------------------
    def _device_epilogue_applies(self, x: NDArray) -> bool:
        """The fused path covers the default normaliser (or none) on signals whose
        channel count equals ``num_outs``; anything else keeps the host epilogue."""
        return (x.ndim in (2, 3) and x.shape[-1] == self.num_outs and x.shape[-2] > 0
                and (self.normalizer is None or self.normalizer is rms_normalize))

    def _decorrelate_on_device(self, x: NDArray, devices=None) -> NDArray:
        stereo_steps = self.mode == LayoutMode.MS or self.width is not None
        if stereo_steps and self.num_outs != 2:
            raise ValueError('Input shape invalid: Expected shape (num samples, 2), '
                             f'but got shape {x.shape[:-1] + (self.num_outs,)}.')
        x = np.ascontiguousarray(x, dtype=np.float32)
        stage = dict(ms_encode=self.mode == LayoutMode.MS, width=self.width,
                     normalize=_normalize_flag(self.normalizer is not None))
        if devices is not None and x.ndim == 3:
            from . import multi
            out = _native.pinned_pool.empty(x.shape[:-1] + (self.num_outs,), np.float32)
            return multi.pool_for(devices).map_streams(self._tap_arrays(), x, out, 'decorrelate', _default_mode, **stage)
        return self._device_table().decorrelate_host(x, _default_mode, **stage)

    def decorrelate_batched(self, input_signals: NDArray, *, devices=None) -> NDArray:
        """``(B, n, num_outs)`` independent signals through the whole stage in one
        device pass (convolution + epilogue on the GPU); float32 result, same shape.
        ``devices='all'`` or a list of device indices: contiguous blocks of the batch on several GPUs
        from this one process (see :func:`convolve_velvet_noise_batched`); the stage's reductions are per
        stream, so nothing crosses devices."""
        x = to_float32(np.asarray(input_signals))
        if x.ndim != 3:
            raise ValueError(f'expected (batch, n, channels), got shape {x.shape}')
        if devices is not None:
            from . import multi
            multi.resolve_devices(devices)              # a bad list is an error whatever path the batch takes
        if _device_epilogue is None and not _use_device_epilogue(self.num_outs, self.normalizer is not None,
                                                                  x.shape[1], True):
            return np.stack([self.decorrelate(sig[:, 0] if x.shape[-1] == 1 else sig) for sig in x]) \
                if len(x) else np.zeros(x.shape[:-1] + (self.num_outs,), np.float32)
        if x.shape[-1] == 1 and self.num_outs == 2 and x.shape[1] > 0 and \
                (self.normalizer is None or self.normalizer is rms_normalize):
            return self._decorrelate_on_device(x, devices if len(x) else None)      # mono signals, fanned out on the device
        if not self._device_epilogue_applies(x):
            return np.stack([self.decorrelate(sig) for sig in x]) if len(x) else np.zeros(x.shape, np.float32)
        return self._decorrelate_on_device(x, devices if len(x) else None)


def decorrelate_bank(input_signal: NDArray, decorrelators: Sequence[VelvetNoise]) -> List[NDArray]:
    """``[d.decorrelate(input_signal) for d in decorrelators]`` with the F convolutions in ONE
    device launch (the tables concatenated channel-wise, the signal fanned out to them) and
    each decorrelator's own epilogue on the host afterwards - the shape of the reference
    optimiser's candidate scan (optimization.py:71, :107-117).  Bit-identical to the loop
    in the exact mode."""
    decorrelators = list(decorrelators)
    if not decorrelators:
        return []
    num_outs = decorrelators[0].num_outs
    if any(d.num_outs != num_outs for d in decorrelators):
        raise ValueError('all decorrelators of a bank must have the same num_outs')
    input_signal = to_float32(input_signal)
    if input_signal.ndim == 1:
        input_signal = mono_to_stereo(input_signal)
    if input_signal.ndim != 2:
        raise IndexError('too many indices for array: decorrelate expects a (n,) or (n, channels) signal')
    x = np.ascontiguousarray(input_signal[:, :num_outs], dtype=np.float32)
    if x.shape[1] != num_outs:
        raise IndexError(f'index {num_outs - 1} is out of bounds for axis 1 with size {input_signal.shape[1]}')
    if x.shape[0] == 0:
        return [d.decorrelate(input_signal) for d in decorrelators]
    arrays = concat_tap_arrays([d._tap_arrays() for d in decorrelators])
    table = _native.TapTable.create(_native.default_context(), arrays.tap_offsets, arrays.tap_index,
                                    arrays.tap_weight, **arrays.kwargs())
    try:
        y = table.convolve_host(x, _default_mode)                           # (n, F * num_outs)
    finally:
        table.close()
    return [d._host_epilogue(input_signal, np.ascontiguousarray(y[:, f * num_outs:(f + 1) * num_outs]))
            for f, d in enumerate(decorrelators)]


# ----------------------------------------------------------------------------
# NumPy-only chain stages (outside the GPU scope)
# ----------------------------------------------------------------------------
@dataclass(kw_only=True)
class HaasEffect(Decorrelator):
    """Delay one channel by ``round(delay_time_seconds * fs)`` samples
    (decorrelation.py:163-230).  Returns float64 ``(n + delay, 2)``."""

    delayed_channel: int = 0
    delay_time_seconds: float = 0.02
    mode: LayoutMode = LayoutMode.LR

    def decorrelate(self, input_signal: NDArray) -> NDArray:
        output_signal = self.haas_delay(to_float32(input_signal))
        if self.width is not None:
            apply_stereo_width(output_signal, self.width)
        return output_signal

    def haas_delay(self, input_signal: NDArray) -> NDArray:
        delay = round(self.delay_time_seconds * self.sample_rate_hz)
        n = len(input_signal)
        mono = input_signal.ndim == 1
        if mono:
            input_signal = mono_to_stereo(input_signal)
        out = np.zeros((n + delay, 2))
        out[:n, :] = input_signal
        mid_side = self.mode == LayoutMode.MS
        if mid_side and not mono:
            LR_to_MS(out)
        out[:, self.delayed_channel] = np.roll(out[:, self.delayed_channel], delay, axis=0)
        if mid_side:
            MS_to_LR(out)
            if mono:
                out *= 0.5          # the duplicated mono channel counted twice
        return out


@dataclass(kw_only=True)
class WhiteNoise(Decorrelator):
    """Dense Gaussian FIR per channel via ``np.convolve(mode='same')``
    (decorrelation.py:670-716) - the comparison baseline of the reference's plots."""

    duration_seconds: float = 0.03
    seed: Optional[int] = None
    white_noise_filter: Any = field(default=None, repr=False)

    def __post_init__(self) -> None:
        rng = np.random.default_rng(self.seed)
        self.white_noise_filter = rng.normal(loc=0, scale=1,
                                             size=(self.fir_length_samples, self.num_outs))

    @property
    def fir_length_samples(self) -> int:
        return int(round(self.sample_rate_hz * self.duration_seconds))

    @property
    def FIR(self) -> NDArray:
        return self.white_noise_filter

    def decorrelate(self, input_signal: NDArray) -> NDArray:
        input_signal = to_float32(input_signal)
        if input_signal.ndim == 1:
            input_signal = mono_to_stereo(input_signal)
        out = np.zeros((len(input_signal), self.num_outs), dtype=np.float32)
        for c in range(self.num_outs):
            out[:, c] = np.convolve(input_signal[:, c], self.white_noise_filter[:, c], mode='same')
        if self.width is not None:
            apply_stereo_width(out, self.width)
        rms_normalize(input_signal, out)
        return out


# ----------------------------------------------------------------------------
# SignalChain   (decorrelation.py:71-153)
# ----------------------------------------------------------------------------
class SignalChain:
    """Fluent, lazily-instantiated cascade of stages; ``chain(x)`` feeds each
    stage the previous stage's output.  ``device_resident=True`` (an extension) keeps the
    signal in HBM between stages that have a device form (``resident.py``)."""

    def __init__(self, *, sample_rate_hz: int, num_outs: int = 2, lazy: bool = True,
                 device_resident: bool = False, _hot: bool = False, _decorrelators=None):
        if _decorrelators is not None:
            raise TypeError(
                'Cannot supply decorrelators directly, use ``SignalChain.velvet_noise``,'
                ' ``SignalChain.haas_effect``, ``SignalChain.white_noise``, or ``SignalChain.stateless``.')
        self.sample_rate_hz = sample_rate_hz
        self.num_outs = num_outs
        self.lazy = lazy
        self.device_resident = bool(device_resident)
        self._resident_pool = None            # device buffers of a resident chain, reused from call to call
        self._hot = bool(_hot) or not lazy
        self._decorrelators: list = []

    def __repr__(self) -> str:
        return (f'SignalChain(sample_rate_hz={self.sample_rate_hz}, num_outs={self.num_outs}, '
                f'lazy={self.lazy}, stages={len(self._decorrelators)})')

    # ---- builders ------------------------------------------------------------
    def velvet_noise(self, **kwargs) -> 'SignalChain':
        return self._add(VelvetNoise, kwargs)

    def haas_effect(self, **kwargs) -> 'SignalChain':
        return self._add(HaasEffect, kwargs)

    def white_noise(self, **kwargs) -> 'SignalChain':
        return self._add(WhiteNoise, kwargs)

    def stateless(self, function: StatelessDecorrelator, *args, **kwargs) -> 'SignalChain':
        """``function`` is later called as ``function(*args, signal, **kwargs)`` -
        ``functools.partial`` semantics, positional extras BEFORE the signal,
        exactly as upstream (decorrelation.py:104-110)."""
        bound = partial(function, *args, **kwargs)
        self._decorrelators.append(bound if self._hot else (lambda: bound))
        return self

    def _add(self, cls, kwargs: dict) -> 'SignalChain':
        rate = kwargs.pop('sample_rate_hz', None)
        if rate is not None and rate != self.sample_rate_hz:
            raise TypeError(f'sample_rate_hz={rate} was supplied to {cls} but differs from the sample '
                            f'rate of the enclosing ``SignalChain`` ({self.sample_rate_hz})')
        params = dict(kwargs)
        params.setdefault('num_outs', self.num_outs)

        def make():
            return cls(sample_rate_hz=self.sample_rate_hz, **params)

        self._decorrelators.append(make() if self._hot else make)
        return self

    # ---- execution -----------------------------------------------------------
    def _init_decorrelators(self) -> None:
        if self._hot:
            return
        self._decorrelators = [factory() for factory in self._decorrelators]
        self._hot = True

    def __call__(self, input_signal: NDArray) -> NDArray:
        self._init_decorrelators()
        if self.device_resident:
            from . import resident
            if self._resident_pool is None:
                self._resident_pool = resident.BufferPool()
            return resident.run(self._decorrelators, input_signal, self._resident_pool)
        signal = input_signal
        for stage in self._decorrelators:
            signal = stage(signal)
        return signal
