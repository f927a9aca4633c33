"""CPU tier of the optimiser row: the oracle's objective against the
scores the reference produced (tests/golden, written by oracle/gen_golden.py), and the
host-side pieces of vndecorrelate_amd.optimization that need no GPU."""
import numpy as np
import pytest

from oracle import vnd_oracle as O
from conftest import make_input
import vndecorrelate_amd.optimization as opt
from vndecorrelate_amd.utils import dsp


def _cases(golden):
    for name, meta in golden.manifest['objective'].items():
        sig = golden.arrays[meta['input']] if isinstance(meta['input'], str) else make_input(meta['input'])
        yield name, meta, sig


def _oracle_output(sig, fs, kappa):
    return O.decorrelate(sig.copy(), sample_rate_hz=fs, duration_seconds=0.03, num_impulses=30,
                         log_distribution_strength=kappa, filtered_channels=(0,), mode='LR', normalize=False,
                         seed=1)


def test_oracle_objective_reproduces_reference_scores(golden):
    for name, meta, sig in _cases(golden):
        want = golden.arrays[f'obj_{name}_scores']
        terms = golden.arrays[f'obj_{name}_terms']
        for k, kappa in enumerate(meta['kappas']):
            out = _oracle_output(sig, meta['sample_rate_hz'], kappa)
            assert O.symmetry_aware_objective(out, **meta['kwargs']) == want[k], (name, kappa)
            got = O.objective_terms(out, angle_limit=meta['kwargs']['angle_limit'])
            assert tuple(float(v) for v in got) == tuple(terms[k]), (name, kappa)
        assert O.local_minima(want, len(want)) == meta['local_minima']


def test_host_objective_pieces_match_reference(golden):
    """polar_coordinates and the moment functions of the drop-in module, on the oracle's output."""
    for name, meta, sig in _cases(golden):
        terms = golden.arrays[f'obj_{name}_terms']
        scores = golden.arrays[f'obj_{name}_scores']
        kw = meta['kwargs']
        for k in (0, 4, 8):
            out = _oracle_output(sig, meta['sample_rate_hz'], meta['kappas'][k])
            radii, thetas, weights = dsp.polar_coordinates(out[:, 0], out[:, 1], normalize=False)
            assert thetas.dtype == np.float32 and np.max(np.abs(thetas)) <= np.float32(np.pi / 2)
            spread = opt.angular_variance(thetas, weights)
            got = (spread, opt.centroid(thetas, weights), opt.polar_skewness(thetas, weights, spread),
                   float(opt.left_right_correlation(out)), opt.max_angular_exceedance(thetas, kw['angle_limit']))
            assert got == tuple(terms[k]), (name, k)
            # the same score from eight float64 moments (what the device returns)
            th, r = thetas.astype(np.float64), radii.astype(np.float64)
            left, right = out[:, 0].astype(np.float64), out[:, 1].astype(np.float64)
            row = [r.sum(), (r * th).sum(), (r * th**2).sum(), (r * th**3).sum(), np.max(np.abs(th)),
                   (left * right).sum(), (left * left).sum(), (right * right).sum()]
            assert abs(opt.score_from_moments(np.array(row), **kw) - scores[k]) <= 2e-4, (name, k)      # float32 sums upstream: ~1e-7 relative
        assert opt.get_local_minima(scores, len(scores)) == meta['local_minima']


def test_polar_coordinates_options():
    left = np.array([1.0, 0.0, -1.0, 0.5, -0.5], np.float32)
    right = np.array([1.0, 1.0, -1.0, -0.5, 0.25], np.float32)
    radii, thetas = dsp.polar_coordinates(left, right, compute_weights=False)
    assert radii.max() <= 1.0 and thetas[0] == 0.0
    _, lr = dsp.polar_coordinates(left, right, mode='LR', semicircular=False, compute_weights=False)
    assert np.allclose(lr, np.arctan2(left, right))
    _, folded, weights = dsp.polar_coordinates(left, right)
    assert np.all(np.abs(folded) <= np.pi / 2 + 1e-6) and abs(weights.sum() - 1) < 1e-6


def test_local_minima_fallback_and_refinement():
    assert opt.get_local_minima(np.array([3.0, 2.0, 1.0, 0.5]), 4) == [3]           # monotone: global minimum
    assert opt.get_local_minima(np.array([3.0, 1.0, 2.0, 0.0, 5.0]), 5) == [1, 3]
    xs = np.linspace(0.0, 1.0, 11)
    best = opt.optimize_local_minima([3, 7], xs, 11, lambda v: (v - 0.33) ** 2 if v < 0.5 else 1 + (v - 0.7) ** 2)
    assert abs(best - 0.33) < 1e-3
