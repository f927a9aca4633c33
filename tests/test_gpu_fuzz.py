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


def _term_scale(arr, x) -> float:
    """max over channels of sum_k |w_k * gain| times max|x|: the size of what is being added up.
    The fma modes round each product differently from mul-then-add, so when the taps cancel
    (output peak << terms; hypothesis finds -x[0] + x[0]) the honest floor is half an ulp of
    the terms, not a fraction of the vanishing peak."""
    w = np.abs(arr.tap_weight.astype(np.float64))
    if arr.seg_offsets is not None and arr.apply_gain and len(w):
        gain = np.zeros(len(w))
        start = 0
        for end, g in zip(arr.seg_end, arr.seg_gain):
            gain[start:end] = abs(float(g))
            start = end
        w = w * gain
    sums = [w[arr.tap_offsets[c]:arr.tap_offsets[c + 1]].sum() for c in range(arr.num_channels)]
    return (max(sums) if sums else 0.0) * (float(np.max(np.abs(x))) if x.size else 0.0)


def _check(ctx, arr, x, want, pairs):
    from vndecorrelate_amd import _native
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight, **arr.kwargs())
    peak = max(float(np.max(np.abs(want))) if want.size else 0.0, 1e-30)
    floor = 2.0 ** -24 * _term_scale(arr, x)
    try:
        ctx.set_variant(pairs)
        for mode in (0, 1, 2):
            y = table.convolve_host(x, mode)
            if mode == 0:
                assert np.array_equal(y, want), f'exact mode, pairs={pairs}'
            else:
                assert np.max(np.abs(y.astype(np.float64) - want)) <= 1e-6 * peak + floor + 1e-30, (mode, pairs)
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
