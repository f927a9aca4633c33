"""CPU oracle for the velvet-noise hot path.  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``vndecorrelate_amd/`` imports it, and the product
path has no CPU fallback.

It restates, in plain NumPy and from the behavioural contract in SURVEY.md §8a,
what the reference (ckonst/VNDecorrelate v1.1.0, pure Python + NumPy) computes
on the hot path.  Citations are ``file:line`` relative to ``/root/reference``.

Parity status: PINNED.  ``oracle/gen_golden.py`` (run once in the build
container, where the reference is mounted) checks every function below
bit-for-bit against the reference itself and against the reference's own
golden files (``audio/viola_decorrelated.wav``, ``audio/vocal_decorrelated.wav``),
then writes the fixtures in ``tests/golden/`` that ``tests/test_oracle_golden.py``
re-checks on every run without the reference present.
"""
from __future__ import annotations

import numpy as np

DEFAULT_ENVELOPE = (0.85, 0.55, 0.35, 0.2)
RMS_EPS = 1e-10


# --------------------------------------------------------------------------
# a3 / a4: tap placement   (src/vndecorrelate/utils/dsp.py:170-201, :204-250)
# --------------------------------------------------------------------------
def log_distribution(strength: float, size: int) -> np.ndarray:
    """``size + 1`` weights ``10^(2*strength*j/size) / (1 + 99*strength)``.

    utils/dsp.py:194-201.  The operation order is kept so the float64 results
    (and therefore the rounded tap positions) are identical.
    """
    ramp = np.arange(size + 1.0) / size
    return (10.0 ** (2.0 * strength * ramp)) / (100.0 * ((1.0 + (strength * 99.0)) / 100.0))


def place_taps(randoms, dist, marks, jitter) -> np.ndarray:
    """Jittered positions ``round(r * max(0, dist*jitter - 1) + marks)`` as int32.

    utils/dsp.py:248-250 (``np.round`` = round-half-even).
    """
    return np.round(randoms * np.fmax(0.0, dist * jitter - 1) + marks).astype(np.int32)


def _draw_placement(seed, num_impulses, num_filters, fir_len, sample_rate_hz,
                    duration_seconds, strength):
    """Common RNG walk of both generators: signs drawn first, then offsets.

    decorrelation.py:573-611 (function) and :488-523 (class).  Returns
    ``(idx int32 (K+1, F), signs float64 (K, F))``.
    """
    rng = np.random.default_rng(seed)
    dist = log_distribution(strength, num_impulses)
    marks = np.cumsum(dist)
    if strength == 0.0:
        marks -= 1.0
    marks *= fir_len / marks[-1]
    u_sign = rng.uniform(low=0, high=1, size=(num_impulses, num_filters))
    u_off = rng.uniform(low=0, high=1, size=(num_impulses + 1, num_filters))
    signs = (2 * np.round(u_sign)) - 1
    jitter = sample_rate_hz / (num_impulses / duration_seconds)
    idx = np.stack(
        [place_taps(u_off[:, f], dist, marks, jitter) for f in range(num_filters)], axis=1
    )
    return idx, signs


def _segment_of(k: int, num_impulses: int, num_segments: int) -> int:
    """decorrelation.py:622 / :540."""
    return int(k / (num_impulses / num_segments))


# --------------------------------------------------------------------------
# a2: dense seeded FIR   (decorrelation.py:549-627)
# --------------------------------------------------------------------------
def generate_velvet_noise(*, duration_seconds, num_impulses, num_outs=2,
                          sample_rate_hz=44100, segment_envelope=DEFAULT_ENVELOPE,
                          log_distribution_strength=1.0, seed=None) -> np.ndarray:
    fir_len = int(duration_seconds * sample_rate_hz)          # truncates (:575)
    fir = np.zeros((fir_len, num_outs), dtype=np.float32)
    env = tuple(segment_envelope) if len(segment_envelope) else (1.0,)
    idx, signs = _draw_placement(seed, num_impulses, num_outs, fir_len,
                                 sample_rate_hz, duration_seconds,
                                 log_distribution_strength)
    for c in range(num_outs):
        for k in range(num_impulses):
            # duplicate positions: last write wins (:623)
            fir[idx[k, c], c] = signs[k, c] * env[_segment_of(k, num_impulses, len(env))]
    return fir


# --------------------------------------------------------------------------
# a1: stateless sparse convolution   (decorrelation.py:630-660)
# --------------------------------------------------------------------------
def convolve_velvet_noise(x: np.ndarray, fir: np.ndarray) -> np.ndarray:
    """``y[n,c] = sum_k w[c,k] * x[n + i[c,k], c]`` over the nonzeros of
    ``fir[:,c]`` in ascending index, accumulated tap by tap into float32."""
    channels = 1 if x.ndim == 1 else x.shape[1]
    if channels > 1 and x.shape[1] != fir.shape[1]:
        raise ValueError(
            f'Input length mismatch: {x.shape[1]} vs {fir.shape[1]} for dimension 1.')
    n = len(x)
    y = np.zeros(x.shape, dtype=np.float32)
    for c in range(channels):
        col = fir[:, c]
        for i in np.flatnonzero(col != 0.0):
            w = col[i]
            stop = n - i if i else n
            y[:max(stop, 0), c] += x[i:, c] * w        # 1-D x -> IndexError, as upstream
    return y


def fir_to_taps(fir: np.ndarray):
    """CSR tap table of a dense FIR: ``offsets[C+1]``, ``idx`` ascending per
    channel, ``w`` in the FIR's dtype.  (np.where order, decorrelation.py:651-654)"""
    fir = np.asarray(fir)
    if fir.ndim == 1:
        fir = fir[:, None]
    offs, idx, w = [0], [], []
    for c in range(fir.shape[1]):
        nz = np.flatnonzero(fir[:, c] != 0.0)
        idx.append(nz.astype(np.int32))
        w.append(fir[nz, c])
        offs.append(offs[-1] + len(nz))
    return (np.asarray(offs, np.int32), np.concatenate(idx) if idx else np.zeros(0, np.int32),
            np.concatenate(w) if w else np.zeros(0, fir.dtype))


def convolve_taps_scalar(x: np.ndarray, offsets, idx, w) -> np.ndarray:
    """Scalar model of a1 for float32 input: per output sample the recurrence
    ``acc = f32(acc + f32(x*w))`` over taps in table order, no FMA.  This is
    the arithmetic the HIP exact mode implements; SURVEY §8a1 probed it
    bit-identical to the slice form above.  Vectorised over n, sequential in k.
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, channels = x.shape
    y = np.zeros((n, channels), np.float32)
    for c in range(channels):
        acc = np.zeros(n, np.float32)
        for k in range(offsets[c], offsets[c + 1]):
            i = int(idx[k])
            if i >= n:
                continue
            term = np.zeros(n, np.float32)
            term[:n - i] = x[i:, c] * np.float32(w[k])
            acc = acc + term
        y[:, c] = acc
    return y


# --------------------------------------------------------------------------
# a7: class-path tap lists   (decorrelation.py:478-546)
# --------------------------------------------------------------------------
def class_fir_length(sample_rate_hz, duration_seconds) -> int:
    return int(round(sample_rate_hz * duration_seconds))       # :452


def generate_class_taps(*, sample_rate_hz, duration_seconds=0.03, num_impulses=30,
                        num_outs=2, segment_envelope=DEFAULT_ENVELOPE,
                        log_distribution_strength=1.0, filtered_channels=(0, 1),
                        seed=None):
    """Per output channel: ``None`` if unfiltered, else a list over segments of
    ``(negatives, positives)`` index lists in generation order; duplicates kept.
    RNG columns are indexed by *output channel number* (:531), so a
    ``filtered_channels`` that is not ``0..F-1`` raises IndexError as upstream.
    """
    env = tuple(segment_envelope) if len(segment_envelope) else (1.0,)
    fir_len = class_fir_length(sample_rate_hz, duration_seconds)
    nfilt = len(filtered_channels)
    idx, signs = _draw_placement(seed, num_impulses, nfilt, fir_len, sample_rate_hz,
                                 duration_seconds, log_distribution_strength)
    out = []
    for c in range(num_outs):
        if c not in filtered_channels:
            out.append(None)
            continue
        col_idx, col_sign = idx[:, c], signs[:, c]             # IndexError if c >= F
        segs = [([], []) for _ in env]
        for k in range(num_impulses):
            s = _segment_of(k, num_impulses, len(env))
            segs[s][int((col_sign[k] + 1) / 2)].append(int(col_idx[k]))
        out.append(segs)
    return out


def class_fir(taps, envelope, fir_len) -> np.ndarray:
    """``VelvetNoise.FIR`` (decorrelation.py:454-472): float64 ``(L, F)``, last
    write wins on duplicates, order = segment, negatives, positives."""
    env = tuple(envelope) if len(envelope) else (1.0,)
    filt = [t for t in taps if t is not None]
    fir = np.zeros((fir_len, len(filt)))
    for f, segs in enumerate(filt):
        for s, (neg, pos) in enumerate(segs):
            for i in neg:
                fir[i, f] = env[s] * -1
            for i in pos:
                fir[i, f] = env[s] * 1
    return fir


# --------------------------------------------------------------------------
# a6: class-path convolution   (decorrelation.py:393-415)
# --------------------------------------------------------------------------
def class_convolve(x: np.ndarray, taps, envelope, num_outs) -> np.ndarray:
    """Per segment: subtract the negative taps, add the positive ones, scale
    by the segment's envelope (skipped for the identity envelope), add into the
    output.  Unfiltered channels are copied through."""
    env = tuple(envelope)
    n = len(x)
    seg = np.zeros(n, dtype=np.float32)
    y = np.zeros((n, num_outs), dtype=np.float32)
    for c in range(num_outs):
        if taps[c] is None:
            y[:, c] = x[:, c]
    for c, segs in enumerate(taps):
        if segs is None:
            continue
        for s, (neg, pos) in enumerate(segs):
            for i in neg:
                seg[:(n - i if i else n) if i < n else 0] -= x[i:, c]
            for i in pos:
                seg[:(n - i if i else n) if i < n else 0] += x[i:, c]
            if env != (1.0,):
                seg *= env[s]
            y[:, c] += seg
            seg.fill(0)
    return y


# --------------------------------------------------------------------------
# a9: epilogue helpers   (utils/dsp.py:21-167)
# --------------------------------------------------------------------------
def _require_stereo(a):
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError(f'Expected shape (num samples, 2), got {a.shape}.')


def lr_to_ms(a):
    _require_stereo(a)
    mid = (a[:, 0] + a[:, 1]) * 0.5
    side = (a[:, 0] - a[:, 1]) * 0.5
    a[:, 0] = mid
    a[:, 1] = side


def ms_to_lr(a):
    _require_stereo(a)
    left = a[:, 0] + a[:, 1]
    right = a[:, 0] - a[:, 1]
    a[:, 0] = left
    a[:, 1] = right


def apply_stereo_width(a, width):
    lr_to_ms(a)
    a[:, 0] *= 1.0 - width
    a[:, 1] *= width
    ms_to_lr(a)


def encode_side(x, y):
    """utils/dsp.py:40-63: mid = x_L + x_R (not halved), side = (y_L - y_R)/2."""
    _require_stereo(x)
    _require_stereo(y)
    mid = x[:, 0] + x[:, 1]
    side = (y[:, 0] - y[:, 1]) * 0.5
    y[:, 0] = (mid + side) * 0.5
    y[:, 1] = (mid - side) * 0.5


def rms_normalize(x, y, eps=RMS_EPS):
    """DUAL_MONO default of utils/dsp.py:87-109: per-channel scale
    ``sqrt(mean(x^2)) / sqrt(mean(y^2) + eps)`` (axis None for 1-D input)."""
    ax_x = None if x.ndim == 1 else 0
    ax_y = None if y.ndim == 1 else 0
    y *= np.sqrt(np.mean(np.square(x), axis=ax_x)) / np.sqrt(
        np.mean(np.square(y), axis=ax_y) + eps)


# --------------------------------------------------------------------------
# a8: VelvetNoise.decorrelate   (decorrelation.py:417-442)
# --------------------------------------------------------------------------
def decorrelate(x, *, sample_rate_hz, num_outs=2, width=None, duration_seconds=0.03,
                num_impulses=30, segment_envelope=DEFAULT_ENVELOPE,
                log_distribution_strength=1.0, normalize=True,
                filtered_channels=(0, 1), mode='MS', seed=None):
    x = x.astype(np.float32, copy=False)
    if x.ndim == 1:
        x = np.column_stack((x, x))
    env = tuple(segment_envelope) if len(segment_envelope) else (1.0,)
    taps = generate_class_taps(
        sample_rate_hz=sample_rate_hz, duration_seconds=duration_seconds,
        num_impulses=num_impulses, num_outs=num_outs, segment_envelope=env,
        log_distribution_strength=log_distribution_strength,
        filtered_channels=filtered_channels, seed=seed)
    y = class_convolve(x, taps, env, num_outs)
    if mode == 'MS':
        encode_side(x, y)
    if width is not None:
        apply_stereo_width(y, width)
    if normalize:
        rms_normalize(x, y)
    return y


def haas_delay_lr(x, *, sample_rate_hz, delay_time_seconds, delayed_channel):
    """LR-mode HaasEffect (decorrelation.py:192-230) — only what is needed to
    replay the reference's committed ``*_decorrelated.wav`` chain."""
    d = round(delay_time_seconds * sample_rate_hz)
    x = x.astype(np.float32, copy=False)
    n = len(x)
    y = np.zeros((n + d, 2))
    y[:n, :] = x if x.ndim == 2 else np.column_stack((x, x))
    y[:, delayed_channel] = np.roll(y[:, delayed_channel], d, axis=0)
    return y


# ---- f3: the optimiser's objective (optimization.py:11-105, utils/dsp.py:374-422) ------------
EPSILON = 1e-10


def polar_coordinates(left, right, *, mode='MS', semicircular=True, normalize=True):
    """utils/dsp.py:374-422: ``(radii, thetas, weights)`` of a stereo signal's samples."""
    thetas = np.arctan2(left - right, left + right) if mode == 'MS' else np.arctan2(left, right)
    if semicircular:
        thetas = np.where(thetas < -np.pi / 2, thetas + np.pi,
                          np.where(thetas > np.pi / 2, thetas - np.pi, thetas))
    radii = np.sqrt(left**2 + right**2)
    if normalize:
        radii /= radii.max() + EPSILON
    weights = radii / (radii.sum() + EPSILON)
    return radii, thetas, weights


def objective_terms(output_signal, *, angle_limit):
    """The five scalars the objective is built from (optimization.py:11-44, :73-99)."""
    _, thetas, weights = polar_coordinates(output_signal[:, 0], output_signal[:, 1], normalize=False)
    spread = float(np.sum(weights * thetas**2))
    mean_theta = float(np.sum(weights * thetas))
    skew = float(np.sum(weights * thetas**3)) / (max(spread, EPSILON) ** 1.5)
    norm_left = np.linalg.norm(output_signal[:, 0]) + EPSILON          # (sic) both channels by ||L||, :14-17
    corr = np.dot(output_signal[:, 0] / norm_left, output_signal[:, 1] / norm_left)
    exceed = max(0.0, float(np.max(np.abs(thetas)) - angle_limit))
    return spread, mean_theta, skew, corr, exceed


def symmetry_aware_objective(output_signal, *, angle_limit, lambda_mean, lambda_skew,
                             lambda_correlation, lambda_penalty):
    """optimization.py:46-105 on an already decorrelated signal; returns the value to minimise."""
    spread, mean_theta, skew, corr, exceed = objective_terms(output_signal, angle_limit=angle_limit)
    objective = (spread - lambda_mean * mean_theta ** 2 - lambda_skew * skew ** 2
                 - lambda_correlation * corr ** 2 - lambda_penalty * exceed ** 2)
    return -objective


def local_minima(scores, grid_size):
    """optimization.py:120-128"""
    found = [i for i in range(1, grid_size - 1) if scores[i] < scores[i - 1] and scores[i] < scores[i + 1]]
    return found if found else [int(np.argmin(scores))]


# ---- f4: HaasEffect (decorrelation.py:192-230), every mode --------------------------------
def haas_effect(x, *, sample_rate_hz, delay_time_seconds=0.02, delayed_channel=0, mode='LR', width=None):
    """float64 ``(n + delay, 2)``: one channel (or the mid / side channel) delayed; the tail of the
    zero-padded buffer wraps to the front (np.roll), i.e. the first ``delay`` samples are silent."""
    d = round(delay_time_seconds * sample_rate_hz)
    x = x.astype(np.float32, copy=False)
    n = len(x)
    mono = x.ndim == 1
    y = np.zeros((n + d, 2))
    y[:n, :] = np.column_stack((x, x)) if mono else x
    if mode == 'MS' and not mono:
        lr_to_ms(y)
    y[:, delayed_channel] = np.roll(y[:, delayed_channel], d, axis=0)
    if mode == 'MS':
        ms_to_lr(y)
        if mono:
            y *= 0.5
    if width is not None:
        apply_stereo_width(y, width)
    return y
