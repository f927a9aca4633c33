"""GPU tier of the chain row: HaasEffect on the device against the
reference's outputs, and device-resident chains against the stage-by-stage chain."""
import hashlib

import numpy as np
import pytest

from conftest import make_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vnd():
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    assert 'gfx950' in ctx.info()['name']
    yield d
    ctx.set_variant(-1)
    d.set_default_mode(d.MODE_EXACT)


def _haas_args(kw):
    return dict(delay=round(kw['delay_time_seconds'] * kw['sample_rate_hz']), delayed_channel=kw['delayed_channel'],
                ms_mode=kw['mode'] == 'MS', width=kw.get('width'))


def test_device_haas_is_bit_identical_to_the_reference(vnd, golden):
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    for name, meta in golden.manifest['haas'].items():
        x = make_input(meta['input'])
        x2 = np.ascontiguousarray(x[:, None] if x.ndim == 1 else x)
        got = _native.haas_host(ctx, x2, **_haas_args(meta['kwargs']))
        assert got.dtype == np.float64 and list(got.shape) == meta['out_shape'], name
        assert hashlib.sha256(got.tobytes()).hexdigest() == meta['out_sha256'], name
    # a batch is the loop
    meta = golden.manifest['haas']['haas_ms_side_width']
    xb = make_input(dict(seed=51, shape=[4, 3001, 2]))
    got = _native.haas_host(ctx, xb, **_haas_args(meta['kwargs']))
    for b in range(4):
        assert np.array_equal(got[b], vnd.HaasEffect(**{k: v for k, v in meta['kwargs'].items()}).decorrelate(xb[b]))
    with pytest.raises(ValueError):
        _native.haas_host(ctx, np.zeros((10, 3), np.float32), delay=1, delayed_channel=0, ms_mode=False, width=None)
    with pytest.raises(ValueError):
        _native.haas_host(ctx, np.zeros((10, 2), np.float32), delay=-1, delayed_channel=0, ms_mode=False, width=None)
    assert _native.haas_host(ctx, np.zeros((0, 2), np.float32), delay=3, delayed_channel=1, ms_mode=True,
                             width=None).shape == (3, 2)


def _chains(vnd, fs, **chain_kw):
    yield 'vn_nonorm>haas_lr', (vnd.SignalChain(sample_rate_hz=fs, **chain_kw)
                                .velvet_noise(duration_seconds=0.02, seed=1, normalizer=None)
                                .haas_effect(delay_time_seconds=0.02, delayed_channel=1, mode='LR'))
    yield 'vn_lr_width>haas_ms', (vnd.SignalChain(sample_rate_hz=fs, **chain_kw)
                                  .velvet_noise(seed=2, normalizer=None, mode='LR', width=0.4)
                                  .haas_effect(delay_time_seconds=0.004, delayed_channel=0, mode='MS', width=0.6))
    yield 'haas>vn_nonorm', (vnd.SignalChain(sample_rate_hz=fs, **chain_kw)
                             .haas_effect(delay_time_seconds=0.01, delayed_channel=1)
                             .velvet_noise(seed=3, normalizer=None))
    yield 'vn>stateless>haas', (vnd.SignalChain(sample_rate_hz=fs, **chain_kw)
                                .velvet_noise(seed=4, normalizer=None)
                                .stateless(np.multiply, 0.5)
                                .haas_effect(delay_time_seconds=0.001))


@pytest.mark.parametrize('shape', [[30011, 2], [30011]])
def test_resident_chain_equals_stage_by_stage(vnd, shape):
    """No normaliser in play: every stage is bit-identical, so the chains are."""
    x = make_input(dict(seed=52, shape=shape))
    for (name, plain), (_, resident) in zip(_chains(vnd, 48000), _chains(vnd, 48000, device_resident=True)):
        want, got = plain(x), resident(x)
        assert got.dtype == want.dtype and got.shape == want.shape, name
        assert np.array_equal(got, want), name
    # int16 audio takes the same float32 cast
    xi = make_input(dict(seed=53, shape=[20000, 2], dist='int16'))
    (name, plain), (_, resident) = next(zip(_chains(vnd, 44100), _chains(vnd, 44100, device_resident=True)))
    assert np.array_equal(resident(xi), plain(xi))


def test_resident_example_chain_with_normaliser(vnd, golden):
    """tests/test_example.py's chain on the viola excerpt.  Exact mode sums the squares in the
    reference's order, so the resident chain is bit-identical normaliser and all; the fast mode keeps that
    order too by default (3e-6 of peak); its fully fused form uses exactly rounded sums where NumPy's
    float32 sum is sequential: 5e-4 of peak."""
    x = golden.arrays['viola_excerpt_in']
    fs = golden.manifest['audio']['viola_excerpt']['fs']

    def build(**kw):
        return (vnd.SignalChain(sample_rate_hz=fs, **kw)
                .velvet_noise(duration_seconds=0.02, num_impulses=30, seed=1)
                .haas_effect(delay_time_seconds=0.02, delayed_channel=1, mode='LR'))

    want, got = build()(x), build(device_resident=True)(x)
    assert got.shape == want.shape and got.dtype == np.float64
    peak = np.max(np.abs(want))
    assert np.array_equal(got, want)
    vnd.set_default_mode(vnd.MODE_FAST)
    try:
        fast = build(device_resident=True)(x)
    finally:
        vnd.set_default_mode(vnd.MODE_EXACT)
    assert np.max(np.abs(fast - want)) <= 3e-6 * peak
    vnd.set_default_mode(vnd.MODE_FAST)
    vnd.set_device_epilogue(True)                      # fully fused, float64 sums: the 5e-4 caveat
    try:
        fused = build(device_resident=True)(x)
    finally:
        vnd.set_default_mode(vnd.MODE_EXACT)
        vnd.set_device_epilogue(None)
    assert np.max(np.abs(fused - want)) <= 5e-4 * peak


def test_stateless_convolve_stage_stays_on_the_device(vnd, golden):
    """The README's stateless stage (decorrelation.py:104-110: ``stateless(convolve_velvet_noise, velvet_noise_filters=fir)``)
    inside a resident chain: velvet noise -> stateless convolution -> Haas without a host round trip between the
    stages, bit-identical to the stage-by-stage chain; the buffers are the chain's own from the second call on."""
    from vndecorrelate_amd import resident
    fs = 48000
    fir = vnd.generate_velvet_noise(duration_seconds=0.02, num_impulses=20, num_outs=2, sample_rate_hz=fs, seed=9)

    def build(**kw):
        return (vnd.SignalChain(sample_rate_hz=fs, **kw)
                .velvet_noise(seed=4, normalizer=None)
                .stateless(vnd.convolve_velvet_noise, velvet_noise_filters=fir)
                .haas_effect(delay_time_seconds=0.001))

    plain, res = build(), build(device_resident=True)
    for seed in (60, 61):
        x = make_input(dict(seed=seed, shape=[30011, 2]))
        want, got = plain(x), res(x)
        assert got.dtype == want.dtype and np.array_equal(got, want)
        assert resident.transfers == {'to_host': 1, 'to_device': 1}, resident.transfers     # in once, out once
    pool = res._resident_pool._buffers
    first = {k: v.data_ptr() for k, v in pool.items()}
    res(make_input(dict(seed=62, shape=[30011, 2])))
    assert {k: v.data_ptr() for k, v in pool.items()} == first                             # nothing reallocated
    # mode keyword, batched form
    chain = (vnd.SignalChain(sample_rate_hz=fs, device_resident=True)
             .stateless(vnd.convolve_velvet_noise_batched, velvet_noise_filters=fir, mode=vnd.MODE_EXACT))
    xb = make_input(dict(seed=63, shape=[3, 5000, 2]))
    assert np.array_equal(chain(xb), vnd.convolve_velvet_noise_batched(xb, fir))
    assert resident.transfers == {'to_host': 1, 'to_device': 1}
    # what has no device form keeps the host function, errors and quirks included: the positional form makes the
    # SIGNAL the filter; a float64 signal multiplies in float64
    quirk_plain = vnd.SignalChain(sample_rate_hz=fs).stateless(vnd.convolve_velvet_noise, fir)
    quirk_res = vnd.SignalChain(sample_rate_hz=fs, device_resident=True).stateless(vnd.convolve_velvet_noise, fir)
    xq = make_input(dict(seed=64, shape=[1000, 2]))
    assert np.array_equal(quirk_res(xq), quirk_plain(xq)) and quirk_res(xq).shape == fir.shape
    x64 = make_input(dict(seed=65, shape=[4000, 2])).astype(np.float64)
    chain64 = vnd.SignalChain(sample_rate_hz=fs, device_resident=True).stateless(vnd.convolve_velvet_noise, velvet_noise_filters=fir)
    assert np.array_equal(chain64(x64), vnd.convolve_velvet_noise(x64, fir))
    with pytest.raises(ValueError):
        vnd.SignalChain(sample_rate_hz=fs, device_resident=True).stateless(
            vnd.convolve_velvet_noise, velvet_noise_filters=fir[:, :1])(xq)
