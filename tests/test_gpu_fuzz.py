"""GPU tier: randomised parity (hypothesis) of the C-ABI convolve against the C oracle -
random channel counts, lengths, tap tables (function- and class-path shapes, duplicates,
pass-through channels, taps beyond the signal), every arithmetic mode."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import c_oracle
from test_properties_cpu import class_table, sparse_fir
from vndecorrelate_amd.taps import class_path_arrays, function_path_arrays

pytestmark = pytest.mark.gpu
SET = settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@pytest.fixture(scope='module')
def ctx():
    from vndecorrelate_amd import _native
    c = _native.default_context()
    yield c
    c.set_variant(-1)


def _check(ctx, arr, x, want, pairs):
    from vndecorrelate_amd import _native
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight, **arr.kwargs())
    peak = max(float(np.max(np.abs(want))) if want.size else 0.0, 1e-30)
    try:
        ctx.set_variant(pairs)
        for mode in (0, 1, 2):
            y = table.convolve_host(x, mode)
            if mode == 0:
                assert np.array_equal(y, want), f'exact mode, pairs={pairs}'
            else:
                assert np.max(np.abs(y.astype(np.float64) - want)) <= 1e-6 * peak + 1e-30, (mode, pairs)
    finally:
        ctx.set_variant(-1)
        table.close()


@SET
@given(fir=sparse_fir(), n=st.integers(1, 6000), batch=st.integers(1, 3), seed=st.integers(0, 2**31 - 1),
       pairs=st.sampled_from([0, 1, 2, 4, 8]))
def test_function_path_tables(ctx, fir, n, batch, seed, pairs):
    x = np.random.default_rng(seed).uniform(-1, 1, (batch, n, fir.shape[1])).astype(np.float32)
    arr = function_path_arrays(fir)
    want = c_oracle.convolve(x, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    _check(ctx, arr, x, want, pairs)


@SET
@given(tab=class_table(), n=st.integers(1, 6000), seed=st.integers(0, 2**31 - 1),
       pairs=st.sampled_from([0, 1, 4]))
def test_class_path_tables(ctx, tab, n, seed, pairs):
    chans, env = tab
    x = np.random.default_rng(seed).uniform(-1, 1, (n, len(chans))).astype(np.float32)
    arr = class_path_arrays(chans, env, env != (1.0,))
    want = c_oracle.convolve(x, arr.tap_offsets, arr.tap_index, arr.tap_weight, seg_off=arr.seg_offsets,
                             seg_end=arr.seg_end, seg_gain=arr.seg_gain, chan_flags=arr.chan_flags,
                             apply_gain=arr.apply_gain)
    _check(ctx, arr, x, want, pairs)
