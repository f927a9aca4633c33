"""GPU parity of the optimiser row: device scan scores against the
scores the reference produced (tests/golden), the moments kernel against float64 NumPy,
and the drivers end to end."""
import contextlib
import io

import numpy as np
import pytest

from conftest import make_input
from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu

SCORE_TOL = 2e-4          # absolute, on scores of ~619: 3e-7 relative (the reference sums in float32)


@pytest.fixture(scope='module')
def vnd():
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    assert 'gfx950' in ctx.info()['name']
    yield d
    ctx.set_variant(-1)
    d.set_default_mode(d.MODE_EXACT)


@pytest.fixture(scope='module')
def opt(vnd):
    import vndecorrelate_amd.optimization as o
    return o


def _quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def _cases(golden):
    for name, meta in golden.manifest['objective'].items():
        sig = golden.arrays[meta['input']] if isinstance(meta['input'], str) else make_input(meta['input'])
        yield name, meta, sig


def _candidates(vnd, meta):
    return [vnd.VelvetNoise(sample_rate_hz=meta['sample_rate_hz'], duration_seconds=0.03, num_impulses=30,
                            log_distribution_strength=k, normalizer=None, filtered_channels=(0,), mode='LR', seed=1)
            for k in meta['kappas']]


def _moments64(y):
    """float32 element maths as NumPy, float64 sums: what the device computes."""
    left, right = y[:, 0], y[:, 1]
    th = np.arctan2(left - right, left + right)
    th = np.where(th < -np.pi / 2, th + np.pi, np.where(th > np.pi / 2, th - np.pi, th))
    r = np.sqrt(left**2 + right**2)
    assert th.dtype == np.float32 and r.dtype == np.float32
    d = np.float64
    return np.array([r.sum(dtype=d), (r * th).sum(dtype=d), (r * th**2).sum(dtype=d), (r * (th**2 * th)).sum(dtype=d),
                     np.max(np.abs(th)) if len(th) else 0.0, (left * right).sum(dtype=d), (left * left).sum(dtype=d),
                     (right * right).sum(dtype=d)])


@pytest.mark.parametrize('mode', ['exact', 'fast'])
def test_grid_scan_matches_reference_scores(vnd, opt, golden, mode):
    vnd.set_default_mode(vnd.MODE_EXACT if mode == 'exact' else vnd.MODE_FAST)
    try:
        for name, meta, sig in _cases(golden):
            want = golden.arrays[f'obj_{name}_scores']
            got = _quiet(opt.grid_scan, sig, _candidates(vnd, meta), **meta['kwargs'])
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= (SCORE_TOL if mode == 'exact' else 5 * SCORE_TOL), (name, got - want)
            if mode == 'exact':
                assert opt.get_local_minima(got, len(got)) == meta['local_minima'], name
    finally:
        vnd.set_default_mode(vnd.MODE_EXACT)


def test_single_objective_is_bit_identical(vnd, opt, golden):
    """One candidate takes the host route on the bit-exact convolution: same float as the reference."""
    for name, meta, sig in _cases(golden):
        want = golden.arrays[f'obj_{name}_scores']
        cands = _candidates(vnd, meta)
        for k in (0, 3, 8):
            assert opt.symmetry_aware_objective(sig, cands[k], **meta['kwargs']) == want[k], (name, k)


def test_optimize_velvet_noise_reproduces_reference(vnd, opt, golden):
    meta = golden.manifest['objective']['viola_excerpt']
    sig = golden.arrays['viola_excerpt_in']
    kappa = _quiet(opt.optimize_velvet_noise, input_signal=sig, sample_rate_hz=44100, duration_seconds=0.03,
                   num_impulses=30, seed=1, grid_size=9)
    assert abs(kappa - meta['optimize_velvet_noise_grid9']) <= 1e-6


@pytest.mark.parametrize('pairs,n', [(1, 1), (1, 777), (3, 5000), (63, 1025), (64, 1025), (100, 4099), (400, 513)])
def test_moments_kernel_against_numpy(vnd, pairs, n):
    """Both kernel shapes (lanes along time for narrow banks, lanes along candidates for wide ones)."""
    import torch
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    y = np.random.default_rng(pairs * 1000 + n).uniform(-1, 1, (n, 2 * pairs)).astype(np.float32)
    y[n // 2] = 0.0                                      # a silent frame: r = 0, theta = 0
    if n > 3:
        y[3, 0::2] = -y[3, 1::2]                         # L + R == 0: theta = +-pi/2 exactly
    yd = torch.from_numpy(y).cuda()
    ws = _native.polar_moments_workspace_bytes(n, pairs)
    work = torch.empty(ws, dtype=torch.uint8, device='cuda')
    out = torch.full((pairs, 8), -1.0, dtype=torch.float64, device='cuda')
    _native.polar_moments_device(ctx, yd.data_ptr(), n, pairs, out.data_ptr(), work.data_ptr(), ws,
                                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for f in range(pairs):
        want = _moments64(y[:, 2 * f:2 * f + 2])
        # atan2f here and NumPy's float32 arctan2 differ by an ulp on some samples, so a sum is
        # compared on the scale of its TERMS (the odd moments cancel): 3e-8 * sum |term|
        scale = _moments64(np.abs(y[:, 2 * f:2 * f + 2]) * np.array([1.0, 0.5], np.float32))
        scale[1:4] = want[0] * np.array([np.pi / 2, (np.pi / 2) ** 2, (np.pi / 2) ** 3])
        assert np.all(np.abs(got[f] - want) <= 3e-8 * np.maximum(scale, 1.0)), (f, got[f], want)
        assert got[f][4] == want[4] or abs(got[f][4] - want[4]) <= 2.4e-7      # max |theta|: one float32 ulp


def test_scan_large_bank_equals_host_loop(vnd, opt):
    """100 candidates x 1 s: device scores vs the host route on each candidate's own output."""
    sig = make_input(dict(seed=47, shape=[48000, 2]))
    kw = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0,
              lambda_penalty=1e3)
    cands = [vnd.VelvetNoise(sample_rate_hz=48000, duration_seconds=0.03, num_impulses=30,
                             log_distribution_strength=k, normalizer=None, filtered_channels=(0,), mode='LR',
                             seed=1 + i % 3) for i, k in enumerate(np.linspace(0, 1, 100))]
    got = _quiet(opt.grid_scan, sig, cands, **kw)
    for i in (0, 17, 50, 99):
        assert abs(got[i] - opt.symmetry_aware_objective(sig, cands[i], **kw)) <= SCORE_TOL, i
    # small launches (sub-banks) give the same numbers
    old = opt._SCAN_BYTES
    opt._SCAN_BYTES = 48000 * 8 * 7
    try:
        again = _quiet(opt.grid_scan, sig, cands, **kw)
    finally:
        opt._SCAN_BYTES = old
    assert np.array_equal(got, again)


def test_mixed_candidates_fall_back_to_the_host(vnd, opt):
    sig = make_input(dict(seed=48, shape=[20000, 2]))
    kw = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0,
              lambda_penalty=1e3)
    cands = [vnd.VelvetNoise(sample_rate_hz=48000, seed=1, normalizer=None, mode='LR', filtered_channels=(0,)),
             vnd.VelvetNoise(sample_rate_hz=48000, seed=1),                                   # MS + normaliser
             vnd.HaasEffect(sample_rate_hz=48000, delay_time_seconds=0.01, mode='LR'),
             vnd.VelvetNoise(sample_rate_hz=48000, seed=2, normalizer=None, mode='LR', width=0.5)]
    got = _quiet(opt.grid_scan, sig, cands, **kw)
    for i, c in enumerate(cands):
        assert abs(got[i] - opt.symmetry_aware_objective(sig, c, **kw)) <= SCORE_TOL, i
    tau = _quiet(opt.optimize_haas_delay, input_signal=sig, sample_rate_hz=48000, max_delay_seconds=0.01, grid_size=5)
    assert 0.0 <= tau <= 0.01


def test_scan_errors(vnd, opt, golden):
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    arr = function_path_arrays(golden.fir('g48k_c3'))                       # 3 channels: not stereo pairs
    table = _native.TapTable.create(_native.default_context(), arr.tap_offsets, arr.tap_index, arr.tap_weight)
    with pytest.raises(ValueError):
        table.scan_host(np.zeros((10, 1), np.float32))
    with pytest.raises(ValueError):
        opt.scan_moments(np.zeros((10, 2, 2), np.float32), [])
