"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the oracle
and against the fixtures captured from the reference.

Bars: VND_MODE_EXACT is BIT-EXACT for float32 input (sha256 of the whole output
equals the reference's); VND_MODE_FMA and non-float32 inputs (rounded to float32
at the boundary) stay within 1e-6 of the output peak - the north-star tolerance.
"""
import json
import pathlib

import os

import numpy as np
import pytest

from conftest import make_input
from oracle import c_oracle
from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu

TOL_PEAK = 1e-6
MANIFEST = json.loads((pathlib.Path(__file__).parent / 'golden' / 'manifest.json').read_text())


@pytest.fixture(scope='module')
def vnd():
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    ctx = _native.default_context()      # raises without the .so or without a GPU
    assert 'gfx950' in ctx.info()['name']
    yield d
    ctx.set_variant(-1)
    d.set_default_mode(d.MODE_EXACT)


def _kw(d):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}


def _fir_for(golden, name, meta):
    if name == 'fn_f64_fir':
        cm = golden.manifest['class_taps'][meta['class']]
        kw = {k: v for k, v in _kw(cm['kwargs']).items() if k not in ('width', 'mode', 'normalizer')}
        return O.class_fir(O.generate_class_taps(num_outs=2, **kw), cm['envelope'], cm['fir_length_samples'])
    return golden.fir(meta['generator'])


# ---- a1: convolve_velvet_noise ------------------------------------------------
@pytest.mark.parametrize('name', sorted(MANIFEST['fn']))
def test_function_path_exact(vnd, golden, name):
    meta = golden.manifest['fn'][name]
    x = make_input(meta['input'])
    fir = _fir_for(golden, name, meta)
    if name == 'fn_cfg4_b4':
        y = vnd.convolve_velvet_noise_batched(x, fir, mode=vnd.MODE_EXACT)
        assert np.array_equal(y[:, :golden.slice], golden.arrays['fn_cfg4_b4_head'])
        assert np.array_equal(y[:, -golden.slice:], golden.arrays['fn_cfg4_b4_tail'])
        import hashlib
        for b, want in enumerate(meta['per_stream_sha256']):
            assert hashlib.sha256(y[b].tobytes()).hexdigest() == want
        return
    y = vnd.convolve_velvet_noise(x, fir, mode=vnd.MODE_EXACT)
    # bit-identical for every operand type: float32 (and the integer types NumPy promotes with a float32
    # weight to float32, e.g. int16 audio) through the float32 kernels, float64 signals or filters through
    # the promoting kernel, which rounds to float32 at every tap as NumPy's  out += x * value  does
    golden.expect(name, y, exact=True)


def test_promoting_path_more_types(vnd, golden):
    """int32 / int64 signals are multiplied in float64 by NumPy too; batched == loop; a float64 filter on a
    float32 signal; and outside exact mode the float32 kernels stay within 1e-6 of peak."""
    fir32 = golden.fir('g48k_k30')
    fir64 = fir32.astype(np.float64) * 1.0000001
    rng = np.random.default_rng(8)
    cases = [(rng.integers(-2**20, 2**20, (5000, 2)).astype(np.int32), fir32),
             (rng.integers(-2**40, 2**40, (3000, 2)), fir32),
             (rng.uniform(-1, 1, (7001, 2)), fir32),
             (rng.uniform(-1, 1, (7001, 2)).astype(np.float32), fir64),
             (rng.uniform(-1, 1, (7001, 2)), fir64)]
    for x, fir in cases:
        want = O.convolve_velvet_noise(x, fir)
        got = vnd.convolve_velvet_noise(x, fir)
        assert got.dtype == np.float32 and np.array_equal(got, want), (x.dtype, fir.dtype)
        # outside exact mode the operands are cast to float32 FIRST (the float32 kernels): the north star's 1e-6 of peak is asserted
        # against the function-path oracle on those float32 operands; against the promoting oracle the cast's own rounding (2^-24
        # relative per operand, ~30 terms) comes on top - a separate, explained bar of 2e-6
        fast = vnd.convolve_velvet_noise(x, fir, mode=vnd.MODE_FAST)
        want32 = O.convolve_velvet_noise(x.astype(np.float32), fir.astype(np.float32))
        assert np.max(np.abs(fast.astype(np.float64) - want32)) <= 1e-6 * np.max(np.abs(want32)), (x.dtype, fir.dtype)
        assert np.max(np.abs(fast.astype(np.float64) - want)) <= 2e-6 * np.max(np.abs(want)), (x.dtype, fir.dtype)
    xb = rng.uniform(-1, 1, (3, 4001, 2))
    yb = vnd.convolve_velvet_noise_batched(xb, fir32)
    for b in range(3):
        assert np.array_equal(yb[b], O.convolve_velvet_noise(xb[b], fir32))


@pytest.mark.parametrize('mode', ['fma', 'fast'])
@pytest.mark.parametrize('name', sorted(n for n in MANIFEST['fn'] if n != 'fn_cfg4_b4'))
def test_function_path_tolerance_modes(vnd, golden, name, mode):
    meta = golden.manifest['fn'][name]
    x = make_input(meta['input'])
    m = vnd.MODE_FMA if mode == 'fma' else vnd.MODE_FAST
    y = vnd.convolve_velvet_noise(x, _fir_for(golden, name, meta), mode=m)
    golden.expect(name, y, exact=False, rtol_peak=TOL_PEAK)


def test_function_path_errors(vnd, golden):
    fir8 = golden.fir('g96k_k64_c8')
    with pytest.raises(ValueError):
        vnd.convolve_velvet_noise(np.zeros((10, 2), np.float32), fir8)
    with pytest.raises(IndexError):
        vnd.convolve_velvet_noise(np.zeros(10, np.float32), golden.fir('g44k_mono'))
    # (n, 1) signal against a 2-column FIR uses column 0, as upstream
    x = make_input(dict(seed=1, shape=[3000, 1]))
    fir = golden.fir('g44k_k30')
    assert np.array_equal(vnd.convolve_velvet_noise(x, fir), O.convolve_velvet_noise(x, fir))


# ---- a6: VelvetNoise.convolve ---------------------------------------------------
@pytest.mark.parametrize('name', sorted(MANIFEST['cls_convolve']))
def test_class_convolve(vnd, golden, name):
    meta = golden.manifest['cls_convolve'][name]
    kw = _kw(golden.manifest['class_taps'][meta['class']]['kwargs'])
    vn = vnd.VelvetNoise(**kw)
    x = make_input(meta['input'])
    y = vn.convolve(x)
    golden.expect(name, y, exact=x.dtype == np.float32, rtol_peak=TOL_PEAK)
    vnd.set_default_mode(vnd.MODE_FMA)       # weights are +-1: FMA is exact here too
    try:
        golden.expect(name, vn.convolve(x), exact=x.dtype == np.float32, rtol_peak=TOL_PEAK)
        vnd.set_default_mode(vnd.MODE_FAST)  # gains folded, free order: tolerance parity
        golden.expect(name, vn.convolve(x), exact=False, rtol_peak=TOL_PEAK)
    finally:
        vnd.set_default_mode(vnd.MODE_EXACT)


# ---- a8: VelvetNoise.decorrelate / SignalChain ----------------------------------
@pytest.mark.parametrize('epilogue', ['default', 'host'])
@pytest.mark.parametrize('name', sorted(MANIFEST['cls_decorrelate']))
def test_class_decorrelate(vnd, golden, name, epilogue):
    """default: the epilogue runs on the device in exact mode (bit-identical, normaliser included);
    host: the NumPy epilogue behind the device convolution.  Both must reproduce the reference."""
    meta = golden.manifest['cls_decorrelate'][name]
    kw = _kw(golden.manifest['class_taps'][meta['class']]['kwargs'])
    vnd.set_device_epilogue(False if epilogue == 'host' else None)
    try:
        y = vnd.VelvetNoise(**kw).decorrelate(make_input(meta['input']))
    finally:
        vnd.set_device_epilogue(None)
    golden.expect(name, y, exact=True)


def test_viola_excerpt_and_chain(vnd, golden):
    a = golden.arrays
    x = a['viola_excerpt_in']
    kw = _kw(golden.manifest['class_taps']['v44k_20ms']['kwargs'])
    vn = vnd.VelvetNoise(**kw)
    assert np.array_equal(vn.decorrelate(x.copy()), a['viola_excerpt_decorrelate'])
    assert np.array_equal(vn.convolve(x), a['viola_excerpt_convolve'])
    assert np.array_equal(vnd.convolve_velvet_noise(x, golden.fir('g44k_20ms')), a['viola_excerpt_fn'])
    # tests/test_example.py's chain: VN (MS) -> Haas (LR)
    rate = golden.manifest['audio']['viola_excerpt']['fs']
    chain = (vnd.SignalChain(sample_rate_hz=rate)
             .velvet_noise(**{k: v for k, v in kw.items() if k != 'sample_rate_hz'})
             .haas_effect(delay_time_seconds=0.02, delayed_channel=1, mode='LR'))
    want = O.haas_delay_lr(a['viola_excerpt_decorrelate'], sample_rate_hz=rate,
                           delay_time_seconds=0.02, delayed_channel=1)
    got = chain(x.copy())
    assert got.dtype == np.float64 and np.array_equal(got, want)


def test_reference_equality_tests(vnd):
    """tests/test_decorrelation.py:71-93 and :172-197 restated on the GPU path."""
    kw = dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=44100,
              segment_envelope=(0.85, 0.55, 0.35, 0.2), log_distribution_strength=1.0, seed=1)
    vn = vnd.VelvetNoise(**kw)
    fir = vnd.generate_velvet_noise(**kw)
    assert np.allclose(vn.FIR, fir, atol=1e-6)
    x = np.random.default_rng(123).random((10000, 2))
    y1 = vnd.convolve_velvet_noise(x, fir)
    y2 = vn.convolve(x)
    assert y1.shape == y2.shape and np.allclose(y1, y2, atol=1e-6)
    chain = (vnd.SignalChain(sample_rate_hz=44100)
             .velvet_noise(duration_seconds=0.03, num_impulses=30, width=0.5)
             .haas_effect(delay_time_seconds=0.0197, delayed_channel=1, mode='LR')
             .haas_effect(delay_time_seconds=0.0096, delayed_channel=1, mode='MS')
             .white_noise(duration_seconds=0.03, width=0.5)
             .stateless(vnd.convolve_velvet_noise,
                        vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30)))
    out = chain(np.zeros(1000))
    assert out.shape[0] > 1000 and out.shape[1] == 2
    out = vnd.VelvetNoise(sample_rate_hz=44100)(np.zeros(1000))
    assert out.shape == (1000, 2)


# ---- kernel variants: every tiling must give the same bits ------------------------
def _variant(pairs=None, cg=0, direct=False):
    v = 0
    if pairs is not None:
        v |= pairs
    v |= cg << 8
    v |= int(direct) << 12
    return v


@pytest.mark.parametrize('channels,gname', [(2, 'g48k_k30'), (1, 'g44k_mono'), (8, 'g96k_k64_c8'),
                                            (3, 'g48k_c3'), (2, 'g48k_k128_u')])
def test_variants_agree(vnd, golden, channels, gname):
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    fir = golden.fir(gname)
    x = make_input(dict(seed=11, shape=[20011, channels]))
    offs, idx, w = O.fir_to_taps(fir)
    want = c_oracle.convolve(x, offs, idx, w)
    cgs = [c for c in (1, 2, 4) if channels % c == 0]
    try:
        for mode in (vnd.MODE_EXACT, vnd.MODE_FMA, vnd.MODE_FAST):
            for direct in (False, True):
                for cg in ([0] if direct else cgs):
                    for r in ([None] if direct else ((1, 2, 3, 4, 6, 8) if mode == vnd.MODE_FAST else (1, 2, 4, 8))):
                        ctx.set_variant(_variant(r, cg, direct))
                        y = vnd.convolve_velvet_noise(x, fir, mode=mode)
                        tag = f'mode={mode} direct={direct} cg={cg} pairs={r}'
                        if mode == vnd.MODE_EXACT:
                            assert np.array_equal(y, want), tag
                        else:
                            err = np.max(np.abs(y.astype(np.float64) - want)) / np.max(np.abs(want))
                            assert err <= TOL_PEAK, (tag, err)
    finally:
        ctx.set_variant(-1)


def test_class_variants_agree(vnd, golden):
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    meta = golden.manifest['cls_convolve']['cls_k128_dups']
    vn = vnd.VelvetNoise(**_kw(golden.manifest['class_taps'][meta['class']]['kwargs']))
    x = make_input(meta['input'])
    try:
        for v in (_variant(1, 1), _variant(4, 2), _variant(8, 1), _variant(direct=True)):
            ctx.set_variant(v)
            golden.expect('cls_k128_dups', vn.convolve(x))
    finally:
        ctx.set_variant(-1)


def test_long_fir_falls_back(vnd):
    """A 2 s FIR cannot be staged in LDS: the direct kernel takes over."""
    fir = vnd.generate_velvet_noise(duration_seconds=2.0, num_impulses=40, sample_rate_hz=48000, seed=4)
    x = make_input(dict(seed=2, shape=[150000, 2]))
    y = vnd.convolve_velvet_noise(x, fir)
    offs, idx, w = O.fir_to_taps(fir)
    assert np.array_equal(y, c_oracle.convolve(x, offs, idx, w, threads=4))


@pytest.mark.parametrize('seconds', [0.1, 0.25, 0.4, 0.7])
def test_mid_length_firs_stay_in_lds(vnd, seconds):
    """FIRs of 0.1-0.7 s need windows of 40-140 KB: the tile shrinks, channels split, and the kernels opt
    in to more than 64 KB of LDS - every mode, the decorrelate stage included."""
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    fir = vnd.generate_velvet_noise(duration_seconds=seconds, num_impulses=40, sample_rate_hz=48000, seed=6)
    x = make_input(dict(seed=9, shape=[2, 90001, 2]))
    arr = function_path_arrays(fir)
    table = _native.TapTable.create(_native.default_context(), arr.tap_offsets, arr.tap_index, arr.tap_weight)
    assert 'direct' not in table.describe(2, 90001, 2, vnd.MODE_FAST)
    want = c_oracle.convolve(x, arr.tap_offsets, arr.tap_index, arr.tap_weight, threads=4)
    assert np.array_equal(table.convolve_host(x, vnd.MODE_EXACT), want)
    peak = np.max(np.abs(want))
    for mode in (vnd.MODE_FMA, vnd.MODE_FAST):
        assert np.max(np.abs(table.convolve_host(x, mode).astype(np.float64) - want)) <= 1e-6 * peak, mode
    from vndecorrelate_amd.utils import dsp
    for mode in (vnd.MODE_EXACT, vnd.MODE_FAST):
        got = table.decorrelate_host(x, mode, ms_encode=True, width=0.4, normalize=_native.NORMALIZE_RMS_REFERENCE_ORDER)
        for b in range(2):
            ref = want[b].copy()
            dsp.encode_signal_to_side_channel(x[b], ref)
            dsp.apply_stereo_width(ref, 0.4)
            dsp.rms_normalize(x[b], ref)
            if mode == vnd.MODE_EXACT:
                assert np.array_equal(got[b], ref), (seconds, b)
            else:
                assert np.max(np.abs(got[b] - ref)) <= 3e-6 * np.max(np.abs(ref)), (seconds, b)


# ---- size-independent properties at BASELINE sizes ----------------------------------
def test_linearity_and_shift_cfg2(vnd, golden):
    fir = golden.fir('g48k_k30')
    rng = np.random.default_rng(5)
    x1 = rng.uniform(-1, 1, (480000, 2)).astype(np.float32)
    # exact dyadic scaling commutes with every float32 op of the path
    y1 = vnd.convolve_velvet_noise(x1, fir)
    y2 = vnd.convolve_velvet_noise(x1 * np.float32(0.25), fir)
    assert np.array_equal(y2, y1 * np.float32(0.25))
    # anti-causal shift: dropping the first s frames of x drops the first s frames of y
    s = 4097
    y3 = vnd.convolve_velvet_noise(np.ascontiguousarray(x1[s:]), fir)
    assert np.array_equal(y3, y1[s:])
    # a unit impulse at frame m reads back the time-reversed filter: y[m - i] = w_i
    m = 300000
    imp = np.zeros((480000, 2), np.float32)
    imp[m] = 1.0
    yi = vnd.convolve_velvet_noise(imp, fir)
    assert np.array_equal(yi[m - fir.shape[0] + 1:m + 1][::-1], fir)
    assert not yi[:m - fir.shape[0] + 1].any() and not yi[m + 1:].any()


def test_batched_equals_loop(vnd, golden):
    fir = golden.fir('g48k_k30')
    x = make_input(dict(seed=3, shape=[37, 48000, 2]))
    yb = vnd.convolve_velvet_noise_batched(x, fir)
    offs, idx, w = O.fir_to_taps(fir)
    assert np.array_equal(yb, c_oracle.convolve(x, offs, idx, w, threads=8))
    # ragged tail: a stream length that is not a multiple of anything
    x = make_input(dict(seed=4, shape=[5, 10007, 2]))
    assert np.array_equal(vnd.convolve_velvet_noise_batched(x, fir),
                          c_oracle.convolve(x, offs, idx, w, threads=4))


# ---- device-pointer entry + table transport -------------------------------------------
def test_device_pointer_api_and_table_roundtrip(vnd, golden):
    import torch
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    ctx = _native.default_context()
    fir = golden.fir('g48k_k30')
    arrays = function_path_arrays(fir)
    table = _native.TapTable.create(ctx, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight)
    image = table.to_bytes()
    clone = _native.TapTable.from_bytes(ctx, image)
    assert clone.to_bytes() == image and clone.total_taps == 60 and clone.num_channels == 2
    x = make_input(dict(seed=8, shape=[6, 30000, 2]))
    xd = torch.from_numpy(x).to('cuda:0')
    yd = torch.empty_like(xd)
    stream = torch.cuda.current_stream().cuda_stream
    clone.convolve_device(xd.data_ptr(), yd.data_ptr(), 6, 30000, 2, vnd.MODE_EXACT, stream)
    torch.cuda.synchronize()
    offs, idx, w = O.fir_to_taps(fir)
    assert np.array_equal(yd.cpu().numpy(), c_oracle.convolve(x, offs, idx, w, threads=4))
    with pytest.raises(ValueError):
        clone.convolve_device(xd.data_ptr(), yd.data_ptr(), 6, 30000, 3, vnd.MODE_EXACT, stream)
    with pytest.raises(ValueError):      # overlapping buffers
        clone.convolve_device(xd.data_ptr(), xd.data_ptr(), 6, 30000, 2, vnd.MODE_EXACT, stream)


@pytest.mark.parametrize('channels,gname', [(2, 'g48k_k30'), (1, 'g44k_mono'), (8, 'g96k_k64_c8'), (3, 'g48k_c3')])
def test_misaligned_device_pointers(vnd, golden, channels, gname):
    """Signals that start 4, 8 or 12 bytes off a 16-byte boundary take the narrower
    buffer-access shapes (kFrame / kDword); every shape must give the same bits."""
    import torch
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    ctx = _native.default_context()
    fir = golden.fir(gname)
    arrays = function_path_arrays(fir)
    table = _native.TapTable.create(ctx, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight)
    offs, idx, w = O.fir_to_taps(fir)
    n, batch = 9001, 3
    x = make_input(dict(seed=21, shape=[batch, n, channels]))
    want = c_oracle.convolve(x, offs, idx, w, threads=4)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        for cg in [c for c in (1, 2, 4) if channels % c == 0]:
            ctx.set_variant(cg << 8)
            for shift in (0, 1, 2, 3):                     # floats
                xin = torch.zeros(x.size + 8, dtype=torch.float32, device='cuda:0')
                yout = torch.full((x.size + 8,), 7.0, dtype=torch.float32, device='cuda:0')
                xin[shift:shift + x.size] = torch.from_numpy(x.ravel()).cuda()
                table.convolve_device(xin.data_ptr() + 4 * shift, yout.data_ptr() + 4 * shift,
                                      batch, n, channels, vnd.MODE_EXACT, stream)
                torch.cuda.synchronize()
                got = yout.cpu().numpy()
                assert np.array_equal(got[shift:shift + x.size].reshape(x.shape), want), (cg, shift)
                # nothing outside the output range was touched
                assert np.all(got[:shift] == 7.0) and np.all(got[shift + x.size:] == 7.0), (cg, shift)
    finally:
        ctx.set_variant(-1)


def test_python_and_c_table_images_agree(vnd, golden):
    """The image that travels between ranks is built in Python on rank 0 and
    consumed by vnd_taps_deserialize: both sides must produce the same bytes."""
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import TapArrays, function_path_arrays
    ctx = _native.default_context()
    fn = function_path_arrays(golden.fir('g96k_k64_c8'))
    t = _native.TapTable.create(ctx, fn.tap_offsets, fn.tap_index, fn.tap_weight)
    assert t.to_bytes() == fn.to_bytes()
    cls = vnd.VelvetNoise(sample_rate_hz=44100, seed=1, filtered_channels=(0,), mode='LR')._tap_arrays()
    t = _native.TapTable.create(ctx, cls.tap_offsets, cls.tap_index, cls.tap_weight, **cls.kwargs())
    assert t.to_bytes() == cls.to_bytes()
    back = TapArrays.from_bytes(t.to_bytes())
    assert np.array_equal(back.seg_end, cls.seg_end) and back.apply_gain == cls.apply_gain


# ---- f1: decorrelate epilogue on the device ----------------------------------------------
def _exact_scale_oracle(x, kw):
    """Reference pipeline with the RMS sums taken in float64 (what exact arithmetic gives);
    the reference's own float32 axis-0 sum is sequential and ~1e-4 off on long signals."""
    kw = dict(kw)
    norm = kw.pop('normalizer', 'default') is not None
    x32 = x.astype(np.float32, copy=False)
    if x32.ndim == 1:
        x32 = np.column_stack((x32, x32))
    y = O.decorrelate(x32.copy(), normalize=False, **kw)
    if norm:
        mx = np.mean(np.square(x32.astype(np.float64)), axis=0).astype(np.float32)
        my = np.mean(np.square(y.astype(np.float64)), axis=0).astype(np.float32)
        y *= (np.sqrt(mx) / np.sqrt(my + np.float32(1e-10))).astype(np.float32)
    return y


@pytest.mark.parametrize('mode', ['exact', 'fast'])
@pytest.mark.parametrize('name', sorted(MANIFEST['cls_decorrelate']))
def test_device_epilogue(vnd, golden, name, mode):
    """exact: separate epilogue passes behind the bit-exact convolution;
    fast: the epilogue fused into the fast kernel's store phase."""
    meta = golden.manifest['cls_decorrelate'][name]
    kw = _kw(golden.manifest['class_taps'][meta['class']]['kwargs'])
    x = make_input(meta['input'])
    vnd.set_device_epilogue(True)
    vnd.set_default_mode(vnd.MODE_EXACT if mode == 'exact' else vnd.MODE_FAST)
    try:
        y = vnd.VelvetNoise(**kw).decorrelate(x.copy())
    finally:
        vnd.set_device_epilogue(None)
        vnd.set_default_mode(vnd.MODE_EXACT)
    ref_meta = meta['out']
    assert list(y.shape) == ref_meta['shape'] and y.dtype == np.float32
    peak = max(ref_meta['max_abs'], 1e-30)
    if kw.get('normalizer', 'default') is None:
        # pointwise steps are bit-identical; the fast convolution stays within 1e-6 of peak
        golden.expect(name, y, exact=mode == 'exact', rtol_peak=TOL_PEAK)
    elif mode == 'exact' and y.shape[-1] >= 2:
        # exact mode sums the squares in the reference's own order (sequential float32): the whole
        # stage, normaliser included, is bit-identical
        golden.expect(name, y, exact=True)
    else:
        golden.expect(name, y, exact=False, rtol_peak=5e-4)      # vs the reference's sequential float32 RMS
        want = _exact_scale_oracle(x, kw)                        # vs exact arithmetic: float32 rounding only
        assert np.max(np.abs(y.astype(np.float64) - want)) / peak <= (2e-6 if mode == 'exact' else 3e-6), name


def test_fused_and_unfused_epilogue_agree(vnd, golden):
    """Same FAST convolution, epilogue fused vs as separate passes (variant bit 24)."""
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    x = make_input(dict(seed=14, shape=[5, 30011, 2]))
    outs = []
    vnd.set_default_mode(vnd.MODE_FAST)
    vnd.set_device_epilogue(True)                        # the fused float64-sum form
    try:
        for variant in (-1, 1 << 24, 2, 8):              # auto (fused), unfused, fused at other tile sizes
            ctx.set_variant(variant)
            vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, width=0.3)
            outs.append(vn.decorrelate_batched(x))
    finally:
        ctx.set_variant(-1)
        vnd.set_default_mode(vnd.MODE_EXACT)
        vnd.set_device_epilogue(None)
    peak = np.max(np.abs(outs[0]))
    for y in outs[1:]:
        assert np.max(np.abs(y - outs[0])) <= 2e-6 * peak


@pytest.mark.parametrize('mode', ['fma', 'fast'])
@pytest.mark.parametrize('name', sorted(MANIFEST['cls_decorrelate']))
def test_default_epilogue_in_tolerance_modes(vnd, golden, name, mode):
    """Default policy outside exact mode: device epilogue with the sums in NumPy's order, so the stage
    differs from the reference only through the convolution's 1e-6 of peak."""
    meta = golden.manifest['cls_decorrelate'][name]
    kw = _kw(golden.manifest['class_taps'][meta['class']]['kwargs'])
    vnd.set_default_mode(vnd.MODE_FMA if mode == 'fma' else vnd.MODE_FAST)
    try:
        y = vnd.VelvetNoise(**kw).decorrelate(make_input(meta['input']))
    finally:
        vnd.set_default_mode(vnd.MODE_EXACT)
    golden.expect(name, y, exact=False, rtol_peak=3e-6)


def test_device_epilogue_batched(vnd, golden):
    kw = _kw(golden.manifest['class_taps']['v44k_width']['kwargs'])
    vn = vnd.VelvetNoise(**kw)
    x = make_input(dict(seed=12, shape=[9, 7001, 2]))
    y = vn.decorrelate_batched(x)
    for b in range(9):
        want = _exact_scale_oracle(x[b], kw)
        assert np.max(np.abs(y[b].astype(np.float64) - want)) <= 2e-6 * np.max(np.abs(want)), b
        # exact mode: the batch is the loop of the (bit-identical) host-epilogue stage
        assert np.array_equal(y[b], vn.decorrelate(x[b])), b
    with pytest.raises(ValueError):
        vnd.set_device_epilogue(True)
        try:
            vnd.VelvetNoise(sample_rate_hz=96000, seed=1, num_impulses=64, num_outs=8,
                            filtered_channels=tuple(range(8))).decorrelate(np.zeros((100, 8), np.float32))
        finally:
            vnd.set_device_epilogue(None)


def test_velvet_noise_regeneration_refreshes_device_table(vnd):
    """Changing a key field regenerates the taps (decorrelation.py:368-379); the cached
    device table must follow, and an envelope edit must take effect at convolve time."""
    x = make_input(dict(seed=31, shape=[6000, 2]))
    vn = vnd.VelvetNoise(sample_rate_hz=44100, seed=1)
    y30 = vn.convolve(x)
    assert np.array_equal(y30, O.class_convolve(x, O.generate_class_taps(sample_rate_hz=44100, seed=1),
                                                (0.85, 0.55, 0.35, 0.2), 2))
    for k in (20, 24, 28, 30, 26):                       # several regenerations: addresses get recycled
        vn.num_impulses = k
        want = O.class_convolve(x, O.generate_class_taps(sample_rate_hz=44100, num_impulses=k, seed=1),
                                (0.85, 0.55, 0.35, 0.2), 2)
        assert np.array_equal(vn.convolve(x), want), k
    vn.segment_envelope = (1.0, 0.5, 0.25, 0.125)        # read at convolve time (decorrelation.py:411-412)
    want = O.class_convolve(x, O.generate_class_taps(sample_rate_hz=44100, num_impulses=26, seed=1),
                            (1.0, 0.5, 0.25, 0.125), 2)
    assert np.array_equal(vn.convolve(x), want)


def test_integration_stub_runs(vnd, golden):
    """The ctypes stub printed in INTEGRATION.md (what a reference maintainer would add) is
    executed as written, against the in-tree library, and must reproduce the reference."""
    import re
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.utils.dsp import check_equal_length
    text = (pathlib.Path(__file__).resolve().parents[1] / 'INTEGRATION.md').read_text()
    block = re.search(r'```python\nimport ctypes, numpy as np\n(.*?)```', text, re.S).group(0)
    code = block.strip('`').replace('python\n', '', 1).replace('"libvnd_amd.so"', repr(str(_native.LIB_PATH)))
    ns = {'check_equal_length': check_equal_length}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    fir = golden.fir('g48k_k30')
    x = make_input(dict(seed=41, shape=[25000, 2]))
    assert np.array_equal(ns['convolve_velvet_noise'](x, fir), O.convolve_velvet_noise(x, fir))
    with pytest.raises(ValueError):
        ns['convolve_velvet_noise'](x, golden.fir('g96k_k64_c8'))


# ---- f1 exact mode: the sequential float32 sum of squares, settled in integers -------------------
def _seq_sums(a):
    """NumPy's axis-0 float32 reduction for (n, C >= 2): a sequential recurrence per channel."""
    with np.errstate(all='ignore'):
        sq = np.square(a)
        return np.cumsum(sq, axis=0, dtype=np.float32)[-1] if len(a) else np.zeros(a.shape[1], np.float32)


def _adversarial_signals():
    rng = np.random.default_rng(77)
    n = 70001
    yield 'int16_valued', rng.integers(-32768, 32767, (n, 2)).astype(np.float32)          # integer squares: ties galore
    yield 'small_integers', rng.integers(0, 4, (n, 2)).astype(np.float32) * (rng.random((n, 2)) < 0.3)
    late = np.zeros((n, 2), np.float32)
    late[9000:] = rng.uniform(-1, 1, (n - 9000, 2))
    yield 'silent_start', late
    yield 'powers_of_two', np.ldexp(1.0, rng.integers(-8, 5, (n, 2))).astype(np.float32)
    yield 'denormal_squares', (rng.uniform(-1, 1, (n, 2)) * 1e-21).astype(np.float32)
    big = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    big[40000, 0] = 3e19                                                                  # square overflows to inf
    big[50000, 1] = np.nan
    yield 'inf_and_nan', big
    yield 'uniform_long', rng.uniform(-1, 1, (480000, 2)).astype(np.float32)
    # audio that came from 16-bit integers: ties in most 2048-frame blocks, several per block (the block-parallel sums
    # settle those blocks from the tally kernel's two parity answers; a binade crossing inside one goes group by group)
    t = np.arange(300000)
    music = np.round(8000 * (np.sin(t * 0.01)[:, None] * np.array([1.0, 0.7]) + 0.3 * rng.standard_normal((300000, 2))))
    yield 'int16_music_scaled', (music / 32768.0).astype(np.float32)
    yield 'odd_multiples_of_a_power_of_two', (rng.integers(0, 8, (150000, 2)) * 2 + 1).astype(np.float32) * np.float32(2.0 ** -7)
    yield 'audio_like', (np.sin(np.arange(200000)[:, None] * np.array([0.01, 0.013])) * 0.2).astype(np.float32)
    yield 'three_channels', rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    yield 'eight_channels_int', rng.integers(-500, 500, (n, 8)).astype(np.float32)
    yield 'sixteen_channels', rng.uniform(-1, 1, (20001, 16)).astype(np.float32)
    yield 'thirty_two_channels_int', rng.integers(-300, 300, (9001, 32)).astype(np.float32)
    yield 'shorter_than_a_group', rng.uniform(-1, 1, (100, 2)).astype(np.float32)
    yield 'one_frame', np.array([[0.5, -0.25]], np.float32)


@pytest.mark.parametrize('name,x', list(_adversarial_signals()), ids=lambda v: v if isinstance(v, str) else '')
def test_exact_rms_sums_are_numpys(vnd, name, x):
    """vnd_decorrelate in exact mode must reproduce NumPy's sequential float32 sums bit for bit - on
    data built to hit the integer fast path's exits (ties, binade crossings, zeros, non-finite)."""
    import torch
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.utils.dsp import rms_normalize
    ctx = _native.default_context()
    n, channels = x.shape
    offsets = np.arange(channels + 1, dtype=np.int32)
    table = _native.TapTable.create(ctx, offsets, np.zeros(channels, np.int32), np.ones(channels, np.float32))
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    ws_bytes = _native.decorrelate_workspace_bytes(1, n, channels)
    ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
    table.decorrelate_device(xd.data_ptr(), yd.data_ptr(), 1, n, channels, mode=vnd.MODE_EXACT, ms_encode=False,
                             width=None, normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes,
                             stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got_sums = ws[:2 * channels].cpu().numpy()             # exact mode: one row of 2C float32 sums (as doubles)
    want = _seq_sums(x)
    assert np.array_equal(got_sums[:channels].astype(np.float32), want, equal_nan=True), (name, got_sums, want)
    assert np.array_equal(got_sums[channels:].astype(np.float32), want, equal_nan=True), name      # y == x here
    ref = x.copy()
    with np.errstate(all='ignore'):
        rms_normalize(x, ref)
    assert np.array_equal(yd.cpu().numpy(), ref, equal_nan=True), name


def _single_channel_signals():
    rng = np.random.default_rng(78)
    for n in (1, 5, 8, 9, 127, 128, 129, 135, 8191, 8192, 8193, 16385, 70001, 480000):
        yield f'uniform_{n}', rng.uniform(-1, 1, (n, 1)).astype(np.float32)
    yield 'int16_valued', rng.integers(-32768, 32767, (50000, 1)).astype(np.float32)
    yield 'quiet', (rng.uniform(-1, 1, (30000, 1)) * 1e-4).astype(np.float32)
    late = np.zeros((40000, 1), np.float32)
    late[9000:] = rng.uniform(-1, 1, (31000, 1))
    yield 'silent_start', late
    bad = rng.uniform(-1, 1, (20000, 1)).astype(np.float32)
    bad[12345, 0] = np.nan
    yield 'nan', bad


@pytest.mark.parametrize('name,x', list(_single_channel_signals()), ids=lambda v: v if isinstance(v, str) else '')
def test_single_channel_sums_are_numpys_pairwise_ones(vnd, name, x):
    """A single-channel table: NumPy sums the (n, 1) arrays pairwise in 8192-element chunks; the device
    (rms_pairwise_kernel) must give the same float32 sums and so the same normalised output, bit for bit,
    for several streams at once (each its own length class: a batch shares n, so the streams differ in data)."""
    import torch
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.utils.dsp import rms_normalize
    ctx = _native.default_context()
    n = x.shape[0]
    table = _native.TapTable.create(ctx, np.arange(2, dtype=np.int32), np.zeros(1, np.int32), np.ones(1, np.float32))
    batch = 3
    xs = np.stack([x, x[::-1].copy(), (x * np.float32(0.37)).astype(np.float32)])
    xd = torch.from_numpy(xs).cuda()
    yd = torch.empty_like(xd)
    ws_bytes = _native.decorrelate_workspace_bytes(batch, n, 1)
    ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
    table.decorrelate_device(xd.data_ptr(), yd.data_ptr(), batch, n, 1, mode=vnd.MODE_EXACT, ms_encode=False,
                             width=None, normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes,
                             stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = ws[:2 * batch].cpu().numpy().astype(np.float32).reshape(batch, 2)       # per stream: sum x^2, sum y^2
    for b in range(batch):
        with np.errstate(all='ignore'):
            want = np.add.reduce(np.square(xs[b]), axis=0)[0]
            ref = xs[b].copy()
            rms_normalize(xs[b], ref)
        assert np.array_equal(got[b], np.array([want, want], np.float32), equal_nan=True), (name, b, got[b], want)
        assert np.array_equal(yd[b].cpu().numpy(), ref, equal_nan=True), (name, b)
    table.close()


def test_single_channel_decorrelate_runs_its_epilogue_on_the_device(vnd, golden):
    """VelvetNoise(num_outs=1) on an (n, 1) signal: the default (device) epilogue equals the reference's
    stored output and the host epilogue, and the launch goes through the pairwise sums (no host fallback)."""
    meta = golden.manifest['cls_decorrelate']['dec_c1_long']
    kw = _kw(golden.manifest['class_taps'][meta['class']]['kwargs'])
    x = make_input(meta['input'])
    vn = vnd.VelvetNoise(**kw)
    assert vnd._use_device_epilogue(1, True, len(x), True) and vn._device_epilogue_applies(x)
    y_dev = vn.decorrelate(x.copy())
    vnd.set_device_epilogue(False)
    try:
        y_host = vn.decorrelate(x.copy())
    finally:
        vnd.set_device_epilogue(None)
    assert y_dev.shape == (len(x), 1) and np.array_equal(y_dev, y_host)
    golden.expect('dec_c1_long', y_dev, exact=True)


def test_block_parallel_sums_equal_the_sequential_kernel(vnd):
    """The block-parallel exact sums (rms_par_*: predicted binades, prefix-summed runs, prefetched crossing
    groups) against the one-workgroup-per-stream kernel and NumPy, on batches whose streams cross binades
    at different places, hit ties, start silent, or carry non-finite samples; forced for a large batch too."""
    import torch
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    rng = np.random.default_rng(123)
    n = 150001
    streams = []
    for k in range(24):
        amp = 10.0 ** rng.uniform(-3, 3)
        sig = (rng.uniform(-1, 1, (n, 2)) * amp).astype(np.float32)
        if k % 4 == 1:
            sig = np.round(sig * 64) / 64                               # short mantissas: ties
        if k % 4 == 2:
            sig[:rng.integers(1, 40000)] = 0                            # silent start
        if k % 6 == 3:
            sig *= np.linspace(0.01, 30, n, dtype=np.float32)[:, None]  # growing level: crossings late in the signal
        if k == 5:
            sig[100000, 0] = np.inf
        if k == 7:
            sig[30000, 1] = np.nan
        streams.append(sig.astype(np.float32))
    x = np.stack(streams)
    offsets = np.arange(3, dtype=np.int32)
    table = _native.TapTable.create(ctx, offsets, np.zeros(2, np.int32), np.ones(2, np.float32))
    st = torch.cuda.current_stream().cuda_stream
    results = {}
    for label, variant, batch in (('parallel', -1, 24), ('sequential', 1 << 19, 24), ('parallel-forced', 1 << 17, 24)):
        ctx.set_variant(variant)
        xd = torch.from_numpy(x[:batch]).cuda()
        yd = torch.empty_like(xd)
        ws_bytes = _native.decorrelate_workspace_bytes(batch, n, 2)
        ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
        table.decorrelate_device(xd.data_ptr(), yd.data_ptr(), batch, n, 2, mode=vnd.MODE_EXACT, ms_encode=False, width=None,
                                 normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
        torch.cuda.synchronize()
        results[label] = (ws[:4 * batch].cpu().numpy().astype(np.float32).reshape(batch, 4), yd.cpu().numpy())
    ctx.set_variant(-1)
    want = np.stack([_seq_sums(s) for s in streams])
    for label, (sums, y) in results.items():
        assert np.array_equal(sums[:, :2], want, equal_nan=True), label
        assert np.array_equal(sums[:, 2:], want, equal_nan=True), label          # y == x for this table
        assert np.array_equal(y, results['sequential'][1], equal_nan=True), label
    table.close()


def test_fortran_ordered_signal_keeps_numpys_sum_order(vnd, golden):
    """NumPy sums a Fortran-ordered (n, 2) signal pairwise per column, not row by row: the default
    policy must then leave the normaliser to NumPy - the result equals the host-epilogue stage."""
    x = np.asfortranarray(make_input(dict(seed=61, shape=[50000, 2])))
    assert not x.flags.c_contiguous
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
    got = vn.decorrelate(x)
    vnd.set_device_epilogue(False)
    try:
        want = vn.decorrelate(x)
    finally:
        vnd.set_device_epilogue(None)
    assert np.array_equal(got, want)
    # and it genuinely differs from the C-ordered sum order
    assert not np.array_equal(got, vn.decorrelate(np.ascontiguousarray(x)))


@pytest.mark.parametrize('num_outs', [1, 3, 16, 40])
def test_default_stage_for_any_channel_count(vnd, num_outs):
    """LR tables of 1 to 40 channels: the default policy (device sums for 2-32 channels, NumPy's otherwise)
    gives the host-epilogue result bit for bit."""
    x = make_input(dict(seed=70 + num_outs, shape=[12001, num_outs]))
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, num_outs=num_outs, filtered_channels=tuple(range(num_outs)),
                         mode='LR', num_impulses=12)
    got = vn.decorrelate(x)
    vnd.set_device_epilogue(False)
    try:
        want = vn.decorrelate(x)
    finally:
        vnd.set_device_epilogue(None)
    assert got.shape == (12001, num_outs) and np.array_equal(got, want)


def test_denormal_range_signal_is_exact(vnd, golden):
    """Products and sums in the float32 denormal range are kept (no flush to zero), as NumPy keeps them."""
    fir = golden.fir('g48k_k30')
    x = (make_input(dict(seed=81, shape=[30011, 2])).astype(np.float64) * 3e-38).astype(np.float32)
    assert np.any((np.abs(x) < np.finfo(np.float32).tiny) & (x != 0))
    want = O.convolve_velvet_noise(x, fir)
    assert np.any((np.abs(want) < np.finfo(np.float32).tiny) & (want != 0))
    assert np.array_equal(vnd.convolve_velvet_noise(x, fir), want)
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, normalizer=None)
    got = vn.decorrelate(x)
    vnd.set_device_epilogue(False)
    try:
        assert np.array_equal(got, vn.decorrelate(x))
    finally:
        vnd.set_device_epilogue(None)


def test_many_short_streams(vnd, golden):
    """3000 streams of 777 frames: shorter than a tile, one partial workgroup per stream and channel group;
    the whole stage (one sums workgroup per stream) equals the loop."""
    from vndecorrelate_amd.utils import dsp
    fir = golden.fir('g48k_k30')
    x = make_input(dict(seed=91, shape=[3000, 777, 2]))
    offs, idx, w = O.fir_to_taps(fir)
    want = c_oracle.convolve(x, offs, idx, w, threads=8)
    assert np.array_equal(vnd.convolve_velvet_noise_batched(x, fir), want)
    fast = vnd.convolve_velvet_noise_batched(x, fir, mode=vnd.MODE_FAST)
    assert np.max(np.abs(fast.astype(np.float64) - want)) <= 1e-6 * np.max(np.abs(want))
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, width=0.25)
    got = vn.decorrelate_batched(x)
    for b in (0, 1, 1499, 2999):
        assert np.array_equal(got[b], vn.decorrelate(x[b])), b
    vnd.set_device_epilogue(False)
    try:
        for b in (7, 2998):
            assert np.array_equal(got[b], vn.decorrelate(x[b])), b           # and the NumPy epilogue agrees
    finally:
        vnd.set_device_epilogue(None)


def test_stage_batches_beyond_the_grid_limit(vnd):
    """70 000 tiny streams: the host layer splits the batch for the epilogue kernels' 16-bit stream index."""
    x = make_input(dict(seed=92, shape=[70000, 40, 2]))
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1, duration_seconds=0.003, num_impulses=8)
    got = vn.decorrelate_batched(x)
    assert got.shape == x.shape
    for b in (0, 32767, 32768, 65535, 65536, 69999):
        assert np.array_equal(got[b], vn.decorrelate(x[b])), b


def test_block_sums_from_the_convolutions_store_phase(vnd):
    """Batches of 65 to 255 streams: the per-table window kernel's store phase leaves the per-block sums of squares of x and of
    the finished y (the predictions the block-parallel exact sums start from), so rms_par_sum_kernel's pass over both arrays is not
    run (decorrelate_dev, EpiFuse::blk_sum).  The stage's sums and output against the one-workgroup-per-stream kernel (variant
    bit 19) bit for bit, streams with ties (16-bit audio), silent starts and growing levels among them, and three streams against
    the oracle's whole stage; a launch the window kernel does not take (unprepared: generic kernel) gives the same bytes."""
    import torch
    from oracle import vnd_oracle as O
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
    rng = np.random.default_rng(321)
    batch, n = 96, 50000 + 37 * 2
    x = rng.uniform(-1, 1, (batch, n, 2)).astype(np.float32)
    x[1] = np.round(x[1] * 32767) / 32768                         # ties in most blocks
    x[2, :20000] = 0                                               # silent start
    x[3] *= np.linspace(0.001, 40, n, dtype=np.float32)[:, None]   # crossings late in the signal
    x[4] *= 1e-4
    st = torch.cuda.current_stream().cuda_stream
    xd = torch.from_numpy(x).cuda()
    ws_bytes = _native.decorrelate_workspace_bytes(batch, n, 2)
    out = {}
    # (the one-workgroup-per-stream kernel itself comes in two shapes: a workgroup per ARRAY of a stream - x's chains, y's chains -
    #  when there are fewer streams than two per CU, one per stream otherwise: VND_EPI_SEQ_SPLIT=0)
    for label, variant, prepare in (('generic', -1, False), ('per-stream', 1 << 19, True), ('per-stream, unsplit', 1 << 19, True),
                                    ('from the store phase', -1, True), ('block sums off', -1, True)):
        table = _native.TapTable.create(ctx, *[getattr(vn._tap_arrays(), k) for k in ('tap_offsets', 'tap_index', 'tap_weight')], **vn._tap_arrays().kwargs())
        ctx.set_variant(variant)
        if prepare:
            table.prepare(batch, n, 2, vnd.MODE_EXACT)
            assert table.describe(batch, n, 2, vnd.MODE_EXACT).startswith('conv_spec_exact_window')
        if label == 'block sums off':
            os.environ['VND_EPI_BLOCK_SUMS'] = '0'
        if label == 'per-stream, unsplit':
            os.environ['VND_EPI_SEQ_SPLIT'] = '0'
        try:
            yd = torch.empty_like(xd)
            ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
            table.decorrelate_device(xd.data_ptr(), yd.data_ptr(), batch, n, 2, mode=vnd.MODE_EXACT, ms_encode=True, width=None,
                                     normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            torch.cuda.synchronize()
        finally:
            os.environ.pop('VND_EPI_BLOCK_SUMS', None)
            os.environ.pop('VND_EPI_SEQ_SPLIT', None)
            ctx.set_variant(-1)
        out[label] = (ws[:4 * batch].cpu().numpy().copy(), yd.cpu().numpy())
        table.close()
    for label, (sums, y) in out.items():
        assert np.array_equal(sums, out['per-stream'][0]), label
        assert np.array_equal(y, out['per-stream'][1]), label
    for b in (0, 1, 3):
        assert np.array_equal(out['from the store phase'][1][b], O.decorrelate(x[b], sample_rate_hz=48000, seed=1)), b


def test_eight_channel_decorrelate_fast_stage_through_the_octet_kernels_sums(vnd, golden):
    """cfg5 through the class API (VelvetNoise(num_outs=8, mode='LR', filtered_channels=0..7).decorrelate, decorrelation.py:417-442):
    in the fast mode the per-channel RMS normaliser's sums of squares leave the octet kernel's store phase and one pass scales.
    Against the golden reference output (dec_c8_lr) and, on a pool at the bench's shape, against the oracle's whole stage; the bar is
    the fused stage's 5e-4 of peak (the reference's RMS is a SEQUENTIAL float32 sum, the device's a float64 one - the exact mode
    keeps the sequential order and is bit-identical: asserted beside it); the scales themselves against float64 sums to 1e-6."""
    import torch
    from vndecorrelate_amd import _native
    from conftest import make_input
    meta = golden.manifest['cls_decorrelate']['dec_c8_lr']
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps'][meta['class']]['kwargs'].items()}
    vn = vnd.VelvetNoise(**kw)
    x = make_input(meta['input'])
    golden.expect('dec_c8_lr', vn.decorrelate(x), exact=True)                      # exact mode: the reference's bits
    ctx = _native.default_context()
    table = vn._device_table()
    ctx.set_variant(1 << 23)                                                       # specialise however little work there is
    vnd.set_default_mode(vnd.MODE_FAST)
    try:
        assert 'pieces=channel-octets' in table.describe(1, len(x), 8, vnd.MODE_FAST)
        golden.expect('dec_c8_lr', vn.decorrelate(x), exact=False, rtol_peak=5e-4)
    finally:
        vnd.set_default_mode(vnd.MODE_EXACT)
        ctx.set_variant(-1)
    # a pool at the bench's shape, ragged in time (the last tile is partial), device resident
    pool, n = 6, 960000 - 1234
    xs = torch.empty((pool, n, 8), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    ys = torch.empty_like(xs)
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 8)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    assert 'pieces=channel-octets' in table.describe(pool, n, 8, vnd.MODE_FAST)
    table.decorrelate_device(xs.data_ptr(), ys.data_ptr(), pool, n, 8, mode=vnd.MODE_FAST, ms_encode=False, width=None, normalize=1,
                             workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
    conv = torch.empty_like(xs)
    table.convolve_device(xs.data_ptr(), conv.data_ptr(), pool, n, 8, vnd.MODE_FAST, st)
    torch.cuda.synchronize()
    # the scale the stage applied, per (stream, channel), against float64 sums of squares of what it scaled
    want_scale = torch.sqrt((xs.double() ** 2).mean(dim=1)) / torch.sqrt((conv.double() ** 2).mean(dim=1) + 1e-10)
    got_scale = (ys.double() * conv.double()).sum(dim=1) / (conv.double() ** 2).sum(dim=1)
    assert float(((got_scale - want_scale).abs() / want_scale).max()) <= 1e-6
    for b in (0, pool - 1):
        want = O.decorrelate(xs[b].cpu().numpy(), **kw)
        err = float(np.max(np.abs(ys[b].cpu().numpy().astype(np.float64) - want)) / np.max(np.abs(want)))
        assert err <= 5e-4, (b, err)
    # ... and the exact stage on the same pool: the octet kernel's store phase leaves the block sums, the tally and stitch follow - NumPy's bits
    assert 'pieces=channel-octets' in table.describe(pool, n, 8, vnd.MODE_EXACT)
    table.decorrelate_device(xs.data_ptr(), ys.data_ptr(), pool, n, 8, mode=vnd.MODE_EXACT, ms_encode=False, width=None, normalize=1,
                             workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
    torch.cuda.synchronize()
    for b in (0, pool - 1):
        assert np.array_equal(ys[b].cpu().numpy(), O.decorrelate(xs[b].cpu().numpy(), **kw)), b


@pytest.mark.parametrize('frames', [200000 + 77, 200000 + 78])      # odd: streams only 8-byte aligned (a pair per workgroup); even: 16-byte accesses (a quad per workgroup where C % 4 == 0)
@pytest.mark.parametrize('channels', [4, 6, 8])
def test_exact_rms_sums_of_wider_signals_block_parallel_equal_the_per_stream_kernel(vnd, channels, frames):
    """The reference-order (NumPy: sequential float32) sums of squares of the exact stage on signals of more than two channels run
    block-parallel, channel pair by channel pair (round 5; before, one workgroup per stream: 8.7 ms for cfg5's pool of 16).  Same
    bits as the per-stream kernel (variant bit 19 keeps it) on every stream of a ragged pool, and as the oracle's whole stage -
    np.mean(np.square(.)) itself - on two of them; signals with exact float32 ties (16-bit audio) and a silent channel included."""
    import torch
    from vndecorrelate_amd import _native
    kw = dict(sample_rate_hz=48000, num_outs=channels, num_impulses=30, filtered_channels=tuple(range(channels)), mode='LR', seed=3)
    vn = vnd.VelvetNoise(**kw)
    table = vn._device_table()
    ctx = _native.default_context()
    pool, n = 5, frames
    rng = np.random.default_rng(channels)
    host = rng.uniform(-1, 1, (pool, n, channels)).astype(np.float32)
    host[1] = (np.round(host[1] * 20000) / 32768.0).astype(np.float32)           # 16-bit audio: squares that tie in float32
    host[2, :, channels - 1] = 0.0                                                # a silent channel: rms 0, scale 0 / sqrt(eps)
    xs = torch.from_numpy(host).cuda()
    ws_bytes = _native.decorrelate_workspace_bytes(pool, n, channels)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    # (forced: the per-table quad / octet kernel takes the small launch - for 4k channels its store phase then leaves the block sums)
    for name, variant in (('block_parallel', -1), ('per_stream', 1 << 19), ('per_table_kernel', 1 << 23)):
        ctx.set_variant(variant)
        try:
            y = torch.empty_like(xs)
            table.decorrelate_device(xs.data_ptr(), y.data_ptr(), pool, n, channels, mode=vnd.MODE_EXACT, ms_encode=False, width=None,
                                     normalize=1, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
            torch.cuda.synchronize()
            out[name] = y.cpu().numpy()
        finally:
            ctx.set_variant(-1)
    assert np.array_equal(out['block_parallel'], out['per_stream'])
    assert np.array_equal(out['per_table_kernel'], out['per_stream'])
    for b in (1, 2):
        assert np.array_equal(out['block_parallel'][b], O.decorrelate(host[b], **kw)), b


def test_fused_fast_stage_over_every_convolution_path(vnd):
    """The fully fused FAST stage (pointwise steps + the normaliser's sums in the convolution's store phase, one scale pass) whichever
    kernel takes the convolution: the window form leaving per-block rows (path 1), the pair-read per-table kernel (path 2: one more
    pass for the sums), the generic fast kernel with its per-tile rows (path 0) - on a pool with a ragged tail (the last tile and the
    last 2048-frame block are partial) and on a mono input fanned out.  Every path: the applied scale within 1e-6 of float64 sums of
    what it scaled, the output within 5e-4 of peak of the oracle's whole stage (sequential float32 RMS) and within 3e-6 of the
    other paths."""
    import torch
    from vndecorrelate_amd import _native
    FORCE, GENERIC, PAIR_READ = 1 << 23, 1 << 25, 1 << 5
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
    table = vn._device_table()
    ctx = _native.default_context()
    st = torch.cuda.current_stream().cuda_stream
    pool, n = 3, 200000 - 124          # (even: the streams of a batch stay 16-byte aligned, which the per-table kernels ask for)
    for cx in (2, 1):
        xs = torch.empty((pool, n, cx), dtype=torch.float32, device='cuda').uniform_(-1, 1)
        ws_bytes = _native.decorrelate_workspace_bytes(pool, n, 2)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
        outs = {}
        for name, variant, starts in (('window', FORCE, 'conv_spec_window'), ('pair_read', FORCE | PAIR_READ, 'conv_spec'), ('generic', GENERIC, 'conv_fast')):
            ctx.set_variant(variant)
            try:
                text = table.describe(pool, n, cx, vnd.MODE_FAST)
                if cx == 2:
                    assert text.startswith(starts) and (name != 'pair_read' or not text.startswith('conv_spec_window')), (name, text)
                y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
                table.decorrelate_device(xs.data_ptr(), y.data_ptr(), pool, n, cx, mode=vnd.MODE_FAST, ms_encode=True, width=0.7, normalize=1,
                                         workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=st)
                torch.cuda.synchronize()
                outs[name] = y.cpu().numpy()
            finally:
                ctx.set_variant(-1)
        host = xs.cpu().numpy()
        for b in (0, pool - 1):
            xin = host[b, :, 0] if cx == 1 else host[b]
            want = O.decorrelate(xin, sample_rate_hz=48000, seed=1, width=0.7)
            peak = float(np.max(np.abs(want)))
            for name, y in outs.items():
                err = float(np.max(np.abs(y[b].astype(np.float64) - want))) / peak
                assert err <= 5e-4, (cx, name, b, err)
        base = outs['generic'].astype(np.float64)
        for name in ('window', 'pair_read'):
            assert float(np.max(np.abs(outs[name] - base))) <= 3e-6 * float(np.max(np.abs(base))), (cx, name)
        # the scale each path applied: out = pointwise(y) * scale, so out_b / out_a is constant per (stream, channel) and the RMS of the output equals
        # the RMS of the input (the normaliser's definition, utils/dsp.py:107-109) to the float32 rounding of the scale
        x2 = np.repeat(host, 2, axis=2) if cx == 1 else host
        for name, y in outs.items():
            rms_in = np.sqrt(np.mean(x2.astype(np.float64) ** 2, axis=1))
            rms_out = np.sqrt(np.mean(y.astype(np.float64) ** 2, axis=1))
            assert np.max(np.abs(rms_out / rms_in - 1.0)) <= 1e-6, (cx, name)
