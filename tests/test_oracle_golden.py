"""CPU tier: the oracle (NumPy and C restatements) against fixtures captured
from the reference itself.  No GPU, no reference checkout needed."""
import numpy as np
import pytest

from conftest import make_input
from oracle import c_oracle
from oracle import vnd_oracle as O

BIG = {'fn_cfg3_uniform', 'fn_cfg3_log', 'fn_cfg5'}   # seconds each in NumPy: C oracle only


def _kw(d):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}


def test_generator_matches_reference_tables(golden):
    for gname, meta in golden.manifest['generators'].items():
        fir = O.generate_velvet_noise(**_kw(meta['kwargs']))
        assert np.array_equal(fir, golden.fir(gname)), gname
        assert list(fir.shape) == meta['fir_shape']


def test_appendix_a_known_answer(golden):
    """44.1 kHz / 30 ms / 30 taps / seed 1."""
    fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2,
                                  sample_rate_hz=44100, seed=1)
    ch0 = [2, 4, 7, 10, 13, 17, 22, 28, 34, 42, 51, 62, 74, 89, 104, 124, 146, 171, 201, 238,
           282, 332, 389, 452, 536, 613, 717, 849, 982, 1158]
    ch1 = [2, 4, 7, 10, 13, 17, 22, 28, 35, 42, 51, 61, 73, 87, 104, 123, 144, 172, 204, 240,
           280, 330, 383, 447, 524, 609, 728, 834, 983, 1150]
    assert np.flatnonzero(fir[:, 0]).tolist() == ch0
    assert np.flatnonzero(fir[:, 1]).tolist() == ch1
    signs0 = '+--+++--' '--+-++-' '+++--+++' '+-+--++'
    assert ''.join('+' if v > 0 else '-' for v in fir[ch0, 0]) == signs0
    mags = np.abs(fir[ch0, 0])
    assert np.array_equal(mags, np.float32([0.85] * 8 + [0.55] * 7 + [0.35] * 8 + [0.2] * 7))


@pytest.mark.parametrize('name', sorted(
    n for n in __import__('json').load(open(__import__('pathlib').Path(__file__).parent
                                             / 'golden' / 'manifest.json'))['fn']))
def test_function_path(golden, name):
    meta = golden.manifest['fn'][name]
    x = make_input(meta['input'])
    if name == 'fn_f64_fir':
        kw = _kw(golden.manifest['class_taps'][meta['class']]['kwargs'])
        taps = O.generate_class_taps(**{k: v for k, v in kw.items()
                                        if k not in ('width', 'mode', 'normalizer')}, num_outs=2)
        fir = O.class_fir(taps, golden.manifest['class_taps'][meta['class']]['envelope'],
                          golden.manifest['class_taps'][meta['class']]['fir_length_samples'])
    else:
        fir = golden.fir(meta['generator'])
    if name == 'fn_cfg4_b4':
        y = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(x.shape[0])])
        golden_head = golden.arrays['fn_cfg4_b4_head']
        assert np.array_equal(y[:, :golden.slice], golden_head)
        assert np.array_equal(y[:, -golden.slice:], golden.arrays['fn_cfg4_b4_tail'])
        yc = c_oracle.convolve(x, *O.fir_to_taps(fir), threads=4)
        assert np.array_equal(y, yc)
        return
    if name not in BIG:
        golden.expect(name, O.convolve_velvet_noise(x, fir))
    if x.dtype == np.float32 and x.size and fir.dtype == np.float32:
        offs, idx, w = O.fir_to_taps(fir)
        golden.expect(name, c_oracle.convolve(x, offs, idx, w, threads=4))


def _class_tables(golden, cname, num_outs):
    from vndecorrelate_amd.taps import class_path_arrays
    meta = golden.manifest['class_taps'][cname]
    env = tuple(meta['envelope'])
    taps = golden.class_taps(cname, num_outs)
    return taps, env, class_path_arrays(taps, env, env != (1.0,))


def test_class_taps_match_reference(golden):
    for cname, meta in golden.manifest['class_taps'].items():
        kw = {k: v for k, v in _kw(meta['kwargs']).items() if k not in ('width', 'mode', 'normalizer')}
        kw.setdefault('num_outs', 2)
        mine = O.generate_class_taps(**kw)
        assert mine == golden.class_taps(cname, kw['num_outs']), cname


@pytest.mark.parametrize('name', ['cls_44k_10k', 'cls_44k_f64', 'cls_cfg2', 'cls_k128_dups',
                                  'cls_k128_u', 'cls_ch0', 'cls_noenv', 'cls_env3', 'cls_c8',
                                  'cls_n_lt_l', 'cls_n1'])
def test_class_convolve(golden, name):
    meta = golden.manifest['cls_convolve'][name]
    cmeta = golden.manifest['class_taps'][meta['class']]
    num_outs = cmeta['kwargs'].get('num_outs', 2)
    taps, env, arrays = _class_tables(golden, meta['class'], num_outs)
    x = make_input(meta['input'])
    golden.expect(name, O.class_convolve(x, taps, env, num_outs))
    if x.dtype == np.float32:
        yc = c_oracle.convolve(x, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight,
                               seg_off=arrays.seg_offsets, seg_end=arrays.seg_end,
                               seg_gain=arrays.seg_gain, chan_flags=arrays.chan_flags,
                               apply_gain=arrays.apply_gain, threads=2)
        golden.expect(name, yc)


@pytest.mark.parametrize('name', ['dec_44k_ms', 'dec_44k_mono', 'dec_44k_i16', 'dec_cfg2', 'dec_20ms',
                                  'dec_width', 'dec_lr_width', 'dec_ch0_lr', 'dec_nonorm',
                                  'dec_c8_lr', 'dec_zeros_mono'])
def test_class_decorrelate(golden, name):
    meta = golden.manifest['cls_decorrelate'][name]
    kw = _kw(golden.manifest['class_taps'][meta['class']]['kwargs'])
    norm = kw.pop('normalizer', 'default')
    golden.expect(name, O.decorrelate(make_input(meta['input']), normalize=norm is not None, **kw))


def test_viola_excerpt(golden):
    a = golden.arrays
    x = a['viola_excerpt_in']
    kw = _kw(golden.manifest['class_taps']['v44k_20ms']['kwargs'])
    assert np.array_equal(O.decorrelate(x.copy(), **kw), a['viola_excerpt_decorrelate'])
    assert np.array_equal(O.convolve_velvet_noise(x, golden.fir('g44k_20ms')), a['viola_excerpt_fn'])


def test_reference_known_answers(golden):
    ka = golden.manifest['known_answers']
    assert ka['density_30ms_30'] == 1000
    assert 818.18 < ka['density_55ms_45'] < 818.19
    assert ka['fir_shape_55ms'] == [2426, 2] and ka['nonzeros_55ms'] == 45
    assert ka['fn_fir_len_55ms'] == 2425          # int() vs int(round()) - SURVEY Appendix B.4
    assert O.class_fir_length(44100, 0.055) == 2426
    for stem in ('viola', 'vocal'):
        assert golden.manifest['audio'][stem]['oracle_reproduces_committed'] is True
