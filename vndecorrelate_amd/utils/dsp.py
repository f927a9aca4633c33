"""Host-side helpers on either side of the velvet-noise hot path.

Same names, arguments and error behaviour as the subset of the reference's
``vndecorrelate.utils.dsp`` that the path uses;
citations are ``file:line`` in ckonst/VNDecorrelate v1.1.0.  These are O(N)
pointwise NumPy steps or O(K) tap-placement maths; the O(N*K) tap sum itself
never runs here - it lives in the HIP extension.

``polar_coordinates`` is here for the optimiser's objective; the
other analysis/plot helpers of the reference (correlograms, sweeps) are out of scope
(SURVEY.md §2 rows 9-10).
"""
from __future__ import annotations

import enum

import numpy as np
from numpy.typing import NDArray

EPSILON: float = 1e-10
IDENTITY_ENVELOPE: tuple = (1.0,)


class _StrEnum(str, enum.Enum):
    """String-valued enum: members compare equal to plain strings ('LR', 'MS')."""

    def __str__(self) -> str:
        return str(self.value)


class NormalizeMode(_StrEnum):          # utils/dsp.py:11-13
    STEREO = 'stereo'
    DUAL_MONO = 'dual_mono'


class LayoutMode(_StrEnum):             # utils/dsp.py:16-18
    LR = 'LR'
    MS = 'MS'


# ---- shape guards (utils/dsp.py:289-310) -------------------------------------
def check_mono(input_signal: NDArray) -> None:
    if input_signal.ndim != 1:
        raise ValueError('Input shape invalid: Expected shape (num samples,), '
                         f'but got shape {input_signal.shape}.')


def check_stereo(input_signal: NDArray) -> None:
    if input_signal.ndim != 2 or input_signal.shape[1] != 2:
        raise ValueError('Input shape invalid: Expected shape (num samples, 2), '
                         f'but got shape {input_signal.shape}.')


def check_equal_length(x: NDArray, y: NDArray, dim: int = 0) -> None:
    if x.shape[dim] != y.shape[dim]:
        raise ValueError('Input length mismatch: Expected signals of equal length, but got lengths '
                         f'{x.shape[dim]} and {y.shape[dim]} for dimension {dim}.')


# ---- layout conversions (utils/dsp.py:66-68, :112-167) -------------------------
def to_float32(input_signal: NDArray) -> NDArray[np.float32]:
    """float32 view of the input; the same object when it already is float32."""
    return input_signal.astype(np.float32, copy=False)


def mono_to_stereo(input_signal: NDArray) -> NDArray:
    check_mono(input_signal)
    return np.column_stack((input_signal, input_signal))


def stereo_to_mono(input_signal: NDArray) -> NDArray:
    check_stereo(input_signal)
    return (input_signal[:, 0] + input_signal[:, 1]) * 0.5


def LR_to_MS(input_signal: NDArray) -> None:
    """In place: channel 0 <- (L+R)/2, channel 1 <- (L-R)/2."""
    check_stereo(input_signal)
    mid = (input_signal[:, 0] + input_signal[:, 1]) * 0.5
    side = (input_signal[:, 0] - input_signal[:, 1]) * 0.5
    input_signal[:, 0] = mid
    input_signal[:, 1] = side


def MS_to_LR(input_signal: NDArray) -> None:
    """In place: channel 0 <- M+S, channel 1 <- M-S."""
    check_stereo(input_signal)
    left = input_signal[:, 0] + input_signal[:, 1]
    right = input_signal[:, 0] - input_signal[:, 1]
    input_signal[:, 0] = left
    input_signal[:, 1] = right


# ---- decorrelate epilogue (utils/dsp.py:21-63, :71-109) ------------------------
def apply_stereo_width(input_signal: NDArray, width: float) -> None:
    """In place: scale mid by (1 - width) and side by width."""
    LR_to_MS(input_signal)
    input_signal[:, 0] *= 1.0 - width
    input_signal[:, 1] *= width
    MS_to_LR(input_signal)


def encode_signal_to_side_channel(input_signal: NDArray, decorrelated_signal: NDArray) -> None:
    """In place on ``decorrelated_signal``: mid = x_L + x_R (not halved),
    side = (y_L - y_R) / 2, then L = (mid+side)/2, R = (mid-side)/2."""
    check_stereo(input_signal)
    check_stereo(decorrelated_signal)
    mid = input_signal[:, 0] + input_signal[:, 1]
    side = (decorrelated_signal[:, 0] - decorrelated_signal[:, 1]) * 0.5
    decorrelated_signal[:, 0] = (mid + side) * 0.5
    decorrelated_signal[:, 1] = (mid - side) * 0.5


def _reduce_axis(signal: NDArray, mode: NormalizeMode):
    return None if (signal.ndim == 1 or mode == NormalizeMode.STEREO) else 0


def peak_normalize(input_signal: NDArray, mode: NormalizeMode = NormalizeMode.DUAL_MONO,
                   epsilon: float = EPSILON) -> None:
    input_signal *= 1.0 / (np.max(np.abs(input_signal), axis=_reduce_axis(input_signal, mode)) + epsilon)


def rms_normalize(input_signal: NDArray, output_signal: NDArray,
                  mode: NormalizeMode = NormalizeMode.DUAL_MONO, epsilon: float = EPSILON) -> None:
    """In place: scale ``output_signal`` to the RMS of ``input_signal``
    (per channel in DUAL_MONO, over everything in STEREO)."""
    rms_in = np.sqrt(np.mean(np.square(input_signal), axis=_reduce_axis(input_signal, mode)))
    rms_out = np.sqrt(np.mean(np.square(output_signal), axis=_reduce_axis(output_signal, mode)) + epsilon)
    output_signal *= rms_in / rms_out


# ---- tap placement (utils/dsp.py:170-286) --------------------------------------
def generate_log_distribution(strength: float, size: int) -> NDArray:
    """``size + 1`` weights ``10**(2*strength*j/size) / (1 + 99*strength)``, j = 0..size."""
    ramp = np.arange(size + 1.0) / size
    return (10.0 ** (2.0 * strength * ramp)) / (100.0 * ((1.0 + (strength * 99.0)) / 100.0))


def apply_log_distribution(randoms: NDArray, log_distribution: NDArray,
                           log_impulse_intervals: NDArray, jitter: float) -> NDArray[np.int32]:
    """Jittered tap positions, rounded half-to-even, as int32."""
    spread = np.fmax(0.0, log_distribution * jitter - 1)
    return np.round(randoms * spread + log_impulse_intervals).astype(np.int32)


def uniform_density(randoms: NDArray, impulse_indexes: NDArray,
                    impulse_interval: float) -> NDArray[np.int32]:
    return np.round(impulse_indexes * impulse_interval + randoms * (impulse_interval - 1)).astype(np.int32)


# ---- polar samples of a stereo signal (utils/dsp.py:374-422) -------------------------
def polar_coordinates(left: NDArray, right: NDArray, mode: LayoutMode = LayoutMode.MS, semicircular: bool = True,
                      normalize: bool = True, compute_weights: bool = True):
    """``(radii, thetas[, weights])`` per sample.  MS mode measures the angle of
    (L - R, L + R); ``semicircular`` folds the inverted-mono half of the circle onto
    [-pi/2, pi/2]; weights are the radii over their sum."""
    if mode == LayoutMode.MS:
        thetas = np.arctan2(left - right, left + right)
    else:
        thetas = np.arctan2(left, right)
    if semicircular:
        below, above = thetas < -np.pi / 2, thetas > np.pi / 2
        thetas = np.where(below, thetas + np.pi, np.where(above, thetas - np.pi, thetas))
    radii = np.sqrt(left**2 + right**2)
    if normalize:
        radii /= radii.max() + EPSILON
    if not compute_weights:
        return radii, thetas
    return radii, thetas, radii / (radii.sum() + EPSILON)
