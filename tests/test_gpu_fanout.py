"""GPU parity of the fan-out entry points (`vnd_*_fanout_*`): a signal with fewer
channels than the tap table, output channel c reading input channel c % in_channels.

The oracle is the plain convolution on the REPLICATED input (that is what the
reference does: `mono_to_stereo` then convolve, decorrelation.py:431-432; or the
optimiser's loop over candidate filters, optimization.py:107-117), so every case
is checked bit for bit in the exact mode and to 1e-6 of peak in the others.
"""
import numpy as np
import pytest

from conftest import make_input
from oracle import c_oracle
from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu

TOL_PEAK = 1e-6


@pytest.fixture(scope='module')
def vnd():
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    assert 'gfx950' in ctx.info()['name']
    yield d
    ctx.set_variant(-1)
    d.set_default_mode(d.MODE_EXACT)
    d.set_device_epilogue(None)


def _table(fir):
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    arrays = function_path_arrays(fir)
    return _native.TapTable.create(_native.default_context(), arrays.tap_offsets, arrays.tap_index,
                                   arrays.tap_weight)


def _replicate(x, channels):
    """(…, n, Cx) -> (…, n, channels) with out[..., c] = x[..., c % Cx]"""
    return np.ascontiguousarray(np.tile(x, (1,) * (x.ndim - 1) + (channels // x.shape[-1],)))


def _close(y, want):
    peak = max(float(np.max(np.abs(want))), 1e-30)
    return float(np.max(np.abs(y.astype(np.float64) - want))) / peak


@pytest.mark.parametrize('n', [1, 7, 1023, 2047, 2049, 30011, 480000])
def test_mono_to_stereo_function_path(vnd, golden, n):
    fir = golden.fir('g48k_k30')                       # (1440, 2)
    offs, idx, w = O.fir_to_taps(fir)
    table = _table(fir)
    x = make_input(dict(seed=40 + n % 7, shape=[n, 1]))
    want = c_oracle.convolve(_replicate(x, 2), offs, idx, w, threads=4)
    assert 'fanout' in table.describe(1, n, 1, vnd.MODE_EXACT)
    y = table.convolve_host(x, vnd.MODE_EXACT)
    assert y.shape == (n, 2) and y.dtype == np.float32
    assert np.array_equal(y, want)
    for mode in (vnd.MODE_FMA, vnd.MODE_FAST):
        assert _close(table.convolve_host(x, mode), want) <= TOL_PEAK, mode


def test_mono_batch_and_every_tile_size(vnd, golden):
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    fir = golden.fir('g48k_k30')
    offs, idx, w = O.fir_to_taps(fir)
    table = _table(fir)
    x = make_input(dict(seed=41, shape=[5, 20011, 1]))
    want = c_oracle.convolve(_replicate(x, 2), offs, idx, w, threads=4)
    try:
        for pairs in (0, 1, 2, 3, 4, 6, 8):
            ctx.set_variant(pairs if pairs else -1)
            for mode in (vnd.MODE_EXACT, vnd.MODE_FAST):
                if mode == vnd.MODE_EXACT and pairs in (3, 6):
                    continue                                   # the ordered kernel has no such tile
                y = table.convolve_host(x, mode)
                if mode == vnd.MODE_EXACT:
                    assert np.array_equal(y, want), pairs
                else:
                    assert _close(y, want) <= TOL_PEAK, pairs
        ctx.set_variant(1 << 8)                                # one channel per workgroup: plain staging, stride 1
        assert 'fanout' not in table.describe(5, 20011, 1, vnd.MODE_EXACT)
        assert np.array_equal(table.convolve_host(x, vnd.MODE_EXACT), want)
        ctx.set_variant(1 << 12)                               # direct kernel
        assert np.array_equal(table.convolve_host(x, vnd.MODE_EXACT), want)
    finally:
        ctx.set_variant(-1)


@pytest.mark.parametrize('in_channels,filters', [(2, 5), (1, 6), (1, 5), (2, 1), (3, 2)])
def test_filter_bank_equals_loop(vnd, in_channels, filters):
    """One signal through F different filters in one launch == F separate convolutions."""
    n = 12345
    x = make_input(dict(seed=43, shape=[n, in_channels]))
    firs = [vnd.generate_velvet_noise(duration_seconds=0.03 - 0.002 * f, num_impulses=30 - f, num_outs=in_channels,
                                      sample_rate_hz=48000, seed=100 + f) for f in range(filters)]
    bank = vnd.convolve_velvet_noise_bank(x, firs, mode=vnd.MODE_EXACT)
    assert bank.shape == (filters, n, in_channels)
    for f, fir in enumerate(firs):
        offs, idx, w = O.fir_to_taps(fir)
        want = c_oracle.convolve(x, offs, idx, w)
        assert np.array_equal(bank[f], want), f
        assert np.array_equal(vnd.convolve_velvet_noise(x, fir, mode=vnd.MODE_EXACT), want), f
    fast = vnd.convolve_velvet_noise_bank(x, firs, mode=vnd.MODE_FAST)
    assert _close(fast, np.asarray(bank, np.float64)) <= TOL_PEAK


def test_decorrelate_bank_equals_loop(vnd):
    """Class-path tables (segments, gains, pass-through channels, duplicates) concatenated."""
    kws = [dict(sample_rate_hz=48000, seed=1),
           dict(sample_rate_hz=48000, seed=2, width=0.4, num_impulses=20),
           dict(sample_rate_hz=48000, seed=3, segment_envelope=(1.0,), duration_seconds=0.02),
           dict(sample_rate_hz=48000, seed=4, filtered_channels=(0,), mode=vnd.LayoutMode.LR),
           dict(sample_rate_hz=48000, seed=5, num_impulses=128, log_distribution_strength=1.0, normalizer=None)]
    for shape in ([9001, 2], [9001]):
        x = make_input(dict(seed=44, shape=shape))
        bank = vnd.decorrelate_bank(x, [vnd.VelvetNoise(**kw) for kw in kws])
        for kw, got in zip(kws, bank):
            want = vnd.VelvetNoise(**kw).decorrelate(x)
            assert got.shape == want.shape and np.array_equal(got, want), kw
    with pytest.raises(ValueError):
        vnd.decorrelate_bank(x, [vnd.VelvetNoise(sample_rate_hz=48000, seed=1),
                                 vnd.VelvetNoise(sample_rate_hz=48000, seed=1, num_outs=4,
                                                 filtered_channels=(0, 1, 2, 3))])


def test_mono_decorrelate_matches_reference_goldens(vnd, golden):
    """The stored reference outputs for mono inputs now come through the fan-out launch."""
    hit = 0
    for name, meta in golden.manifest['cls_decorrelate'].items():
        if len(meta['input']['shape']) != 1:
            continue
        kw = {k: (tuple(v) if isinstance(v, list) else v)
              for k, v in golden.manifest['class_taps'][meta['class']]['kwargs'].items()}
        x = make_input(meta['input'])
        golden.expect(name, vnd.VelvetNoise(**kw).decorrelate(x), exact=x.dtype == np.float32, rtol_peak=TOL_PEAK)
        hit += 1
    assert hit >= 1


@pytest.mark.parametrize('mode', ['exact', 'fast'])
def test_mono_device_epilogue(vnd, mode):
    """Whole stage on the device from a mono signal: equals the stage on the duplicated signal."""
    x = make_input(dict(seed=45, shape=[3, 50021, 1]))
    vnd.set_default_mode(vnd.MODE_EXACT if mode == 'exact' else vnd.MODE_FAST)
    try:
        for kw in (dict(sample_rate_hz=48000, seed=1), dict(sample_rate_hz=48000, seed=2, width=0.3),
                   dict(sample_rate_hz=48000, seed=3, normalizer=None)):
            vn = vnd.VelvetNoise(**kw)
            got = vn.decorrelate_batched(x)
            want = vn.decorrelate_batched(_replicate(x, 2))
            assert got.shape == want.shape == (3, 50021, 2)
            if mode == 'exact':
                assert np.array_equal(got, want), kw
            else:
                assert _close(got, want.astype(np.float64)) <= 2e-6, kw
        vnd.set_device_epilogue(True)
        vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
        got = vn.decorrelate(x[0, :, 0])
        want = vn.decorrelate(_replicate(x[0], 2))
        assert (np.array_equal(got, want) if mode == 'exact' else _close(got, want.astype(np.float64)) <= 2e-6)
    finally:
        vnd.set_device_epilogue(None)
        vnd.set_default_mode(vnd.MODE_EXACT)


def test_fanout_device_pointers_misaligned(vnd, golden):
    import torch
    fir = golden.fir('g48k_k30')
    offs, idx, w = O.fir_to_taps(fir)
    table = _table(fir)
    n, batch = 9001, 3
    x = make_input(dict(seed=46, shape=[batch, n, 1]))
    want = c_oracle.convolve(_replicate(x, 2), offs, idx, w, threads=4)
    stream = torch.cuda.current_stream().cuda_stream
    for shift in (0, 1, 2, 3):
        xin = torch.zeros(x.size + 8, dtype=torch.float32, device='cuda:0')
        yout = torch.full((want.size + 8,), 7.0, dtype=torch.float32, device='cuda:0')
        xin[shift:shift + x.size] = torch.from_numpy(x.ravel()).cuda()
        for mode in (vnd.MODE_EXACT, vnd.MODE_FAST):
            table.convolve_device(xin.data_ptr() + 4 * shift, yout.data_ptr() + 4 * shift, batch, n, 1, mode, stream)
            torch.cuda.synchronize()
            got = yout.cpu().numpy()
            body = got[shift:shift + want.size].reshape(want.shape)
            assert (np.array_equal(body, want) if mode == vnd.MODE_EXACT else _close(body, want) <= TOL_PEAK), shift
            assert np.all(got[:shift] == 7.0) and np.all(got[shift + want.size:] == 7.0), shift


def test_fanout_errors(vnd, golden):
    table = _table(golden.fir('g96k_k64_c8'))            # 8 channels
    with pytest.raises(ValueError):
        table.convolve_host(np.zeros((100, 3), np.float32))          # 3 does not divide 8
    y = table.convolve_host(np.zeros((100, 4), np.float32))          # 4 does
    assert y.shape == (100, 8) and not y.any()
    assert table.convolve_host(np.zeros((0, 2), np.float32)).shape == (0, 8)
