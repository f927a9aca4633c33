"""Parameter search of the reference's ``vndecorrelate.optimization``.

Same functions, keyword arguments and return values as
``src/vndecorrelate/optimization.py``: a scalar objective built from the
amplitude-weighted angular moments of a decorrelated signal's polar samples
(:46-105), a grid scan over candidate decorrelators (:107-117), local minima of the
scan (:120-128) refined with SciPy's bounded scalar minimiser (:131-157), and the two
drivers ``optimize_velvet_noise`` (:230-310) and ``optimize_haas_delay`` (:160-227).

What moves to the GPU is the scan.  The reference decorrelates and scores the F
candidates one after the other; here the candidates' tap tables are concatenated into
one bank, the signal is uploaded once, ONE fan-out launch convolves all of them and a
reduction kernel turns the result into eight moments per candidate
(``vnd_scan_bank_f32_host``), so F x 64 bytes come back instead of F signals.  That
covers candidates that are plain velvet-noise convolutions - ``VelvetNoise`` in LR
mode without width or normaliser, which is what ``optimize_velvet_noise`` builds
(:259-271); anything else (MS encode, width, normalisers, ``HaasEffect``) is scored on
the host from its ``decorrelate`` output exactly as upstream.  A single
``symmetry_aware_objective`` call always takes the host route, so its value is
bit-identical to the reference's given the bit-identical exact-mode convolution;
scanned scores agree to ~1e-7 relative (float64 sums where NumPy adds float32).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
from numpy.typing import NDArray

from . import _native
from . import decorrelation as _dec
from .decorrelation import Decorrelator, HaasEffect, VelvetNoise
from .taps import class_path_bank_arrays
from .utils.dsp import EPSILON, LayoutMode, polar_coordinates, to_float32

# one bank's device output is n * 2F floats: keep it under this many bytes per launch
_SCAN_BYTES = 2 << 30


# ---- the objective's terms (optimization.py:11-44) -----------------------------------
def left_right_correlation(stereo_signal: NDArray) -> float:
    """Dot product of the channels, both scaled by the LEFT channel's norm (as upstream, :14-17)."""
    left_norm = np.linalg.norm(stereo_signal[:, 0]) + EPSILON
    return np.dot(stereo_signal[:, 0] / left_norm, stereo_signal[:, 1] / left_norm)


def angular_variance(thetas: NDArray, weights: NDArray) -> float:
    return float(np.sum(weights * thetas**2))


def centroid(thetas: NDArray, weights: NDArray) -> float:
    return float(np.sum(weights * thetas))


def polar_skewness(thetas: NDArray, weights: NDArray, angular_variance: float) -> float:
    return float(np.sum(weights * thetas**3)) / (max(angular_variance, EPSILON) ** 1.5)


def max_angular_exceedance(thetas: NDArray, angle_limit: float) -> float:
    return max(0.0, float(np.max(np.abs(thetas)) - angle_limit))


def _combine(spread: float, mean_theta: float, skew: float, correlation: float, exceedance: float, *,
             lambda_mean: float, lambda_skew: float, lambda_correlation: float, lambda_penalty: float) -> float:
    """optimization.py:79-105: maximise spread, penalise the rest; minimiser convention."""
    objective = (spread
                 - lambda_mean * mean_theta ** 2
                 - lambda_skew * skew ** 2
                 - lambda_correlation * correlation ** 2
                 - lambda_penalty * exceedance ** 2)
    return -objective


def symmetry_aware_objective(input_signal: NDArray, decorrelator: Decorrelator, *, angle_limit: float,
                             lambda_mean: float, lambda_skew: float, lambda_correlation: float,
                             lambda_penalty: float) -> float:
    """Score of one decorrelator on one signal (optimization.py:46-105); lower is better."""
    output_signal = decorrelator.decorrelate(input_signal)
    _, thetas, weights = polar_coordinates(output_signal[:, 0], output_signal[:, 1], normalize=False)
    spread = angular_variance(thetas, weights)
    return _combine(spread, centroid(thetas, weights), polar_skewness(thetas, weights, spread),
                    left_right_correlation(output_signal), max_angular_exceedance(thetas, angle_limit),
                    lambda_mean=lambda_mean, lambda_skew=lambda_skew, lambda_correlation=lambda_correlation,
                    lambda_penalty=lambda_penalty)


# ---- the scan -------------------------------------------------------------------------
def _scannable(decorrelator) -> bool:
    """A candidate whose ``decorrelate`` is the bare convolution of a stereo pair."""
    return (isinstance(decorrelator, VelvetNoise) and decorrelator.num_outs == 2
            and decorrelator.mode == LayoutMode.LR and decorrelator.width is None
            and not decorrelator.normalizer)


def score_from_moments(moments: NDArray, *, angle_limit: float, lambda_mean: float, lambda_skew: float,
                       lambda_correlation: float, lambda_penalty: float) -> float:
    """The objective from one row of device moments
    ``{sum r, sum r*t, sum r*t^2, sum r*t^3, max|t|, sum L*R, sum L^2, sum R^2}``."""
    s0, s1, s2, s3, tmax, lr, ll, _ = (float(v) for v in moments)
    total = s0 + EPSILON                                 # weights = radii / (radii.sum() + EPSILON)
    spread = s2 / total
    skew = (s3 / total) / (max(spread, EPSILON) ** 1.5)
    correlation = lr / (np.sqrt(ll) + EPSILON) ** 2
    return _combine(spread, s1 / total, skew, correlation, max(0.0, tmax - angle_limit),
                    lambda_mean=lambda_mean, lambda_skew=lambda_skew, lambda_correlation=lambda_correlation,
                    lambda_penalty=lambda_penalty)


def scores_from_moments(moments: NDArray, *, angle_limit: float, lambda_mean: float, lambda_skew: float,
                        lambda_correlation: float, lambda_penalty: float) -> NDArray:
    """:func:`score_from_moments` for all rows of an ``(F, 8)`` moments matrix at once: the same float64
    operations element by element, without F trips through the interpreter (NumPy's array ``**`` and
    Python's float ``**`` may differ in the last bit: 6e-16 relative on the scores)."""
    m = np.asarray(moments, np.float64)
    s0, s1, s2, s3, tmax, lr, ll = (m[:, k] for k in range(7))
    total = s0 + EPSILON
    spread = s2 / total
    skew = (s3 / total) / (np.maximum(spread, EPSILON) ** 1.5)
    correlation = lr / (np.sqrt(ll) + EPSILON) ** 2
    exceedance = np.maximum(0.0, tmax - angle_limit)
    objective = (spread - lambda_mean * (s1 / total) ** 2 - lambda_skew * skew ** 2
                 - lambda_correlation * correlation ** 2 - lambda_penalty * exceedance ** 2)
    return -objective


def scan_moments(input_signal: NDArray, decorrelators: Sequence[VelvetNoise], *,
                 mode: Optional[int] = None) -> NDArray:
    """``(F, 8)`` float64 device moments of ``d.decorrelate(input_signal)`` for scannable
    candidates: one upload, one fan-out convolution per sub-bank, only the moments return."""
    x = to_float32(np.asarray(input_signal))
    if x.ndim == 1:
        x = x[:, None]                                   # the device fans the one channel out (mono_to_stereo)
    elif x.ndim != 2 or x.shape[1] < 2:
        raise ValueError(f'expected a mono (n,) or stereo (n, 2) signal, got shape {x.shape}')
    else:
        x = x[:, :2]
    x = np.ascontiguousarray(x, dtype=np.float32)
    frames = max(x.shape[0], 1)
    per_launch = max(1, _SCAN_BYTES // (frames * 8))
    mode = _dec._default_mode if mode is None else mode
    rows: List[NDArray] = []
    for first in range(0, len(decorrelators), per_launch):
        arrays = class_path_bank_arrays([d._tap_member() for d in decorrelators[first:first + per_launch]])
        table = _native.TapTable.create(_native.default_context(), arrays.tap_offsets, arrays.tap_index,
                                        arrays.tap_weight, **arrays.kwargs())
        try:
            rows.append(table.scan_host(x, mode))
        finally:
            table.close()
    return np.concatenate(rows) if rows else np.zeros((0, _native.MOMENTS))


def grid_scan(input_signal: NDArray, decorrelators: Sequence[Decorrelator], **kwargs) -> NDArray:
    """Scores of every candidate (optimization.py:107-117).  Velvet-noise candidates without an
    epilogue are scored on the device in one pass; the rest one by one on the host."""
    print('Starting Grid Scan')
    decorrelators = list(decorrelators)
    scores = np.empty(len(decorrelators), np.float64)
    on_device = [i for i, d in enumerate(decorrelators) if _scannable(d)]
    if on_device and np.asarray(input_signal).shape[0] > 0:
        moments = scan_moments(input_signal, [decorrelators[i] for i in on_device])
        scores[on_device] = scores_from_moments(moments, **kwargs)
    else:
        on_device = []
    for i in sorted(set(range(len(decorrelators))) - set(on_device)):
        scores[i] = symmetry_aware_objective(input_signal, decorrelators[i], **kwargs)
    return scores


def get_local_minima(scores: NDArray, grid_size: int) -> List[int]:
    """Interior grid points lower than both neighbours, else the global minimum (:120-128)."""
    inner = [i for i in range(1, grid_size - 1) if scores[i - 1] > scores[i] < scores[i + 1]]
    return inner if inner else [int(np.argmin(scores))]


def optimize_local_minima(local_minima: List[int], scalars: NDArray, grid_size: int,
                          scalar_objective: Callable[[float], float]):
    """Bounded scalar minimisation between the neighbours of each local minimum (:131-157)."""
    from scipy.optimize import minimize_scalar
    best_scalar, best_score = 0.0, np.inf
    print('Starting Local Minima optimization')
    for i in local_minima:
        bounds = (scalars[max(0, i - 1)], scalars[min(grid_size - 1, i + 1)])
        result = minimize_scalar(fun=scalar_objective, bounds=bounds, method='bounded', options={'xatol': 1e-4})
        if result.fun < best_score:
            best_score, best_scalar = result.fun, result.x
    return best_scalar


def _search(input_signal: NDArray, scalars: NDArray, make: Callable[[float], Decorrelator], grid_size: int,
            weights: dict):
    scores = grid_scan(input_signal, [make(value) for value in scalars], **weights)
    return optimize_local_minima(get_local_minima(scores, grid_size), scalars, grid_size,
                                 lambda value: symmetry_aware_objective(input_signal, make(value), **weights))


def optimize_haas_delay(*, input_signal: NDArray, sample_rate_hz: int, max_delay_seconds: int,
                        grid_size: int = 400, angle_limit: float = np.pi / 4, lambda_mean: float = 5.0,
                        lambda_skew: float = 2.0, lambda_correlation: float = 15.0,
                        lambda_penalty: float = 1e3) -> float:
    """Best ``delay_time_seconds`` in ``[0, max_delay_seconds]`` for an LR ``HaasEffect`` (:160-227)."""
    weights = dict(angle_limit=angle_limit, lambda_mean=lambda_mean, lambda_skew=lambda_skew,
                   lambda_correlation=lambda_correlation, lambda_penalty=lambda_penalty)
    return _search(input_signal, np.linspace(0.0, max_delay_seconds, grid_size),
                   lambda tau: HaasEffect(sample_rate_hz=sample_rate_hz, delay_time_seconds=tau, mode='LR'),
                   grid_size, weights)


def optimize_velvet_noise(*, input_signal: NDArray, sample_rate_hz: int, duration_seconds: float,
                          num_impulses: int, seed: int = 1, grid_size: int = 400,
                          angle_limit: float = np.pi / 4, lambda_mean: float = 5.0, lambda_skew: float = 2.0,
                          lambda_correlation: float = 15.0, lambda_penalty: float = 1e3) -> float:
    """Best ``log_distribution_strength`` in ``[0, 1]`` for a one-sided LR ``VelvetNoise`` (:230-310)."""
    weights = dict(angle_limit=angle_limit, lambda_mean=lambda_mean, lambda_skew=lambda_skew,
                   lambda_correlation=lambda_correlation, lambda_penalty=lambda_penalty)

    def make(kappa: float) -> VelvetNoise:
        return VelvetNoise(sample_rate_hz=sample_rate_hz, duration_seconds=duration_seconds,
                           num_impulses=num_impulses, log_distribution_strength=kappa, normalizer=None,
                           filtered_channels=(0,), mode='LR', seed=seed)

    return _search(input_signal, np.linspace(0.0, 1.0, grid_size), make, grid_size, weights)
