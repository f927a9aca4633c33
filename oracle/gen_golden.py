#!/usr/bin/env python3
"""Generate ``tests/golden/`` from the real reference.  BUILD-CONTAINER ONLY.

TEST INFRASTRUCTURE.  Runs where ``/root/reference`` is mounted (it never
travels to the GPU box).  The reference needs Python >= 3.12 syntax; under the
3.10 interpreter of this image it is loaded from its unmodified source text
with the three-item in-memory shim recorded in SURVEY.md §8c (``enum.StrEnum``,
``typing.Self``, and the PEP 695 ``type X = ...`` line read as an assignment).

What it does, for every case below:
  1. runs the reference function (``convolve_velvet_noise``,
     ``generate_velvet_noise``, ``VelvetNoise.convolve`` / ``.decorrelate`` /
     ``.FIR``) on a seeded input,
  2. runs the restatement in ``oracle/vnd_oracle.py`` on the same input and
     requires BIT-IDENTICAL output (this is what pins the oracle),
  3. stores the input recipe (seed/shape/dtype, or the samples themselves when
     they are real audio) and the reference output: whole if small, otherwise
     head + tail slices, sha256 of the float32 bytes and max|y|.
It also replays ``tests/test_example.py``'s chain on viola/vocal and checks the
reference's committed ``audio/*_decorrelated.wav`` bit-for-bit, against both
the reference code and the oracle.

Usage:  python oracle/gen_golden.py        (writes tests/golden/*.npz + manifest.json)
"""
from __future__ import annotations

import enum
import hashlib
import json
import pathlib
import sys
import types
import typing

import numpy as np

REPO = pathlib.Path(__file__).resolve().parents[1]
REF = pathlib.Path('/root/reference')
OUT = REPO / 'tests' / 'golden'
SLICE = 2048
sys.path.insert(0, str(REPO))

from oracle import vnd_oracle as O  # noqa: E402


def load_reference():
    if not hasattr(enum, 'StrEnum'):
        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)
        enum.StrEnum = StrEnum
    if not hasattr(typing, 'Self'):
        typing.Self = typing.TypeVar('Self')
    src = REF / 'src' / 'vndecorrelate'

    def load(name, path, patch=lambda t: t):
        m = types.ModuleType(name)
        m.__file__ = str(path)
        if path.name == '__init__.py':
            m.__path__ = [str(path.parent)]
        sys.modules[name] = m
        exec(compile(patch(path.read_text()), str(path), 'exec'), m.__dict__)
        return m

    load('vndecorrelate', src / '__init__.py')
    load('vndecorrelate.utils', src / 'utils' / '__init__.py')
    dsp = load('vndecorrelate.utils.dsp', src / 'utils' / 'dsp.py')
    dec = load('vndecorrelate.decorrelation', src / 'decorrelation.py',
               lambda t: t.replace('type _LazyDecorrelator = ', '_LazyDecorrelator = '))
    opt = load('vndecorrelate.optimization', src / 'optimization.py')
    return dsp, dec, opt


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_input(spec) -> np.ndarray:
    """The input recipe every test re-creates on its own box."""
    rng = np.random.default_rng(spec['seed'])
    shape = tuple(spec['shape'])
    kind = spec.get('dist', 'uniform_pm1')
    if kind == 'uniform_pm1':
        x = rng.uniform(-1, 1, shape)
    elif kind == 'uniform_01':
        x = rng.uniform(0, 1, shape)
    elif kind == 'int16':
        return rng.integers(-32768, 32767, shape, dtype=np.int16)
    elif kind == 'zeros':
        x = np.zeros(shape)
    elif kind == 'impulse':
        x = np.zeros(shape)
        x[spec['at']] = 1.0
    else:
        raise ValueError(kind)
    return x.astype(spec.get('dtype', 'float32'))


def pack_output(y: np.ndarray) -> dict:
    y = np.ascontiguousarray(y)
    d = {'shape': list(y.shape), 'dtype': str(y.dtype), 'sha256': sha(y),
         'max_abs': float(np.max(np.abs(y))) if y.size else 0.0}
    return d


def store_arrays(y: np.ndarray, prefix: str, arrays: dict):
    if len(y) <= 2 * SLICE:
        arrays[prefix + '_full'] = y
    else:
        arrays[prefix + '_head'] = y[:SLICE].copy()
        arrays[prefix + '_tail'] = y[-SLICE:].copy()


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype or not np.array_equal(a, b, equal_nan=True):
        raise SystemExit(f'ORACLE MISMATCH in {what}: shapes {a.shape}/{b.shape} '
                         f'dtypes {a.dtype}/{b.dtype} '
                         f'maxdiff {np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) if a.shape == b.shape and a.size else "n/a"}')


# ---------------------------------------------------------------------------
GEN = {   # generator kwargs by name
    'g44k_k30':   dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=44100, seed=1),
    'g44k_20ms':  dict(duration_seconds=0.02, num_impulses=30, num_outs=2, sample_rate_hz=44100, seed=1),
    'g48k_k30':   dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1),
    'g48k_k128_u': dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                        log_distribution_strength=0.0, seed=1),
    'g48k_k128_h': dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                        log_distribution_strength=0.5, seed=1),
    'g48k_k128_l': dict(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                        log_distribution_strength=1.0, seed=1),
    'g96k_k64_c8': dict(duration_seconds=0.03, num_impulses=64, num_outs=8, sample_rate_hz=96000, seed=1),
    'g44k_mono':  dict(duration_seconds=0.03, num_impulses=30, num_outs=1, sample_rate_hz=44100, seed=7),
    'g44k_noenv': dict(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=44100,
                       segment_envelope=(), seed=3),
    'g44k_env3':  dict(duration_seconds=0.5, num_impulses=15, num_outs=2, sample_rate_hz=44100,
                       segment_envelope=(1.0, 0.5, 0.25), seed=5),
    'g44k_55ms':  dict(duration_seconds=0.055, num_impulses=45, num_outs=2, sample_rate_hz=44100, seed=11),
    'g48k_c3':    dict(duration_seconds=0.01, num_impulses=12, num_outs=3, sample_rate_hz=48000,
                       log_distribution_strength=0.3, seed=9),
}

FN_CASES = [  # (name, generator, input spec)
    ('fn_44k_k30_10k',  'g44k_k30',   dict(seed=0, shape=[10000, 2])),
    ('fn_44k_k30_f64',  'g44k_k30',   dict(seed=2, shape=[10000, 2], dist='uniform_01', dtype='float64')),
    ('fn_44k_k30_i16',  'g44k_k30',   dict(seed=3, shape=[6000, 2], dist='int16')),
    ('fn_cfg2',         'g48k_k30',   dict(seed=0, shape=[480000, 2])),
    ('fn_cfg3_uniform', 'g48k_k128_u', dict(seed=0, shape=[2880000, 2])),
    ('fn_cfg3_log',     'g48k_k128_l', dict(seed=0, shape=[2880000, 2])),
    ('fn_k128_half',    'g48k_k128_h', dict(seed=4, shape=[50000, 2])),
    ('fn_cfg5',         'g96k_k64_c8', dict(seed=0, shape=[960000, 8])),
    ('fn_20ms',         'g44k_20ms',  dict(seed=0, shape=[44100, 2])),
    ('fn_mono_col',     'g44k_mono',  dict(seed=0, shape=[5000, 1])),
    ('fn_noenv',        'g44k_noenv', dict(seed=0, shape=[4000, 2])),
    ('fn_long_fir',     'g44k_env3',  dict(seed=0, shape=[30000, 2])),
    ('fn_c3',           'g48k_c3',    dict(seed=0, shape=[3001, 3])),
    ('fn_n1',           'g44k_k30',   dict(seed=0, shape=[1, 2])),
    ('fn_n0',           'g44k_k30',   dict(seed=0, shape=[0, 2])),
    ('fn_n_lt_l',       'g44k_k30',   dict(seed=0, shape=[100, 2])),
    ('fn_n_eq_l',       'g44k_k30',   dict(seed=0, shape=[1323, 2])),
    ('fn_odd_len',      'g48k_k30',   dict(seed=5, shape=[4099, 2])),
    ('fn_tap0',         'g48k_k128_u', dict(seed=6, shape=[3000, 2])),
    ('fn_impulse',      'g44k_k30',   dict(seed=0, shape=[2000, 2], dist='impulse', at=1500)),
]

CLS = {  # VelvetNoise kwargs by name
    'v44k':       dict(sample_rate_hz=44100, duration_seconds=0.03, num_impulses=30, seed=1),
    'v44k_20ms':  dict(sample_rate_hz=44100, duration_seconds=0.02, num_impulses=30, seed=1,
                       log_distribution_strength=1.0, mode='MS', filtered_channels=(0, 1)),
    'v48k':       dict(sample_rate_hz=48000, duration_seconds=0.03, num_impulses=30, seed=1),
    'v48k_k128_l': dict(sample_rate_hz=48000, duration_seconds=0.03, num_impulses=128, seed=1),
    'v48k_k128_u': dict(sample_rate_hz=48000, duration_seconds=0.03, num_impulses=128, seed=1,
                        log_distribution_strength=0.0),
    'v44k_ch0_lr': dict(sample_rate_hz=44100, seed=1, filtered_channels=(0,), mode='LR'),
    'v44k_noenv': dict(sample_rate_hz=44100, seed=3, segment_envelope=()),
    'v44k_env3':  dict(sample_rate_hz=44100, seed=5, num_impulses=15, duration_seconds=0.5,
                       segment_envelope=(1.0, 0.5, 0.25)),
    'v44k_width': dict(sample_rate_hz=44100, seed=1, width=0.5),
    'v44k_lr_w':  dict(sample_rate_hz=44100, seed=2, width=0.25, mode='LR'),
    'v44k_nonorm': dict(sample_rate_hz=44100, seed=1, normalizer=None),
    'v96k_c8':    dict(sample_rate_hz=96000, seed=1, num_impulses=64, num_outs=8, mode='LR',
                       filtered_channels=tuple(range(8))),
    'v44k_55ms':  dict(sample_rate_hz=44100, duration_seconds=0.055, num_impulses=45, seed=11),
    'v44k_c1':    dict(sample_rate_hz=44100, seed=1, num_outs=1, filtered_channels=(0,), mode='LR'),
}

CLS_CONV_CASES = [  # VelvetNoise.convolve
    ('cls_44k_10k',   'v44k',        dict(seed=0, shape=[10000, 2])),
    ('cls_44k_f64',   'v44k',        dict(seed=2, shape=[10000, 2], dist='uniform_01', dtype='float64')),
    ('cls_cfg2',      'v48k',        dict(seed=0, shape=[480000, 2])),
    ('cls_k128_dups', 'v48k_k128_l', dict(seed=0, shape=[100000, 2])),
    ('cls_k128_u',    'v48k_k128_u', dict(seed=0, shape=[100000, 2])),
    ('cls_ch0',       'v44k_ch0_lr', dict(seed=0, shape=[8000, 2])),
    ('cls_noenv',     'v44k_noenv',  dict(seed=0, shape=[8000, 2])),
    ('cls_env3',      'v44k_env3',   dict(seed=0, shape=[30000, 2])),
    ('cls_c8',        'v96k_c8',     dict(seed=0, shape=[20000, 8])),
    ('cls_n_lt_l',    'v44k',        dict(seed=0, shape=[100, 2])),
    ('cls_n1',        'v44k',        dict(seed=0, shape=[1, 2])),
]

CLS_DEC_CASES = [  # VelvetNoise.decorrelate
    ('dec_44k_ms',     'v44k',        dict(seed=0, shape=[20000, 2])),
    ('dec_44k_mono',   'v44k',        dict(seed=1, shape=[20000])),
    ('dec_44k_i16',    'v44k',        dict(seed=3, shape=[9000, 2], dist='int16')),
    ('dec_cfg2',       'v48k',        dict(seed=0, shape=[480000, 2])),
    ('dec_20ms',       'v44k_20ms',   dict(seed=0, shape=[44100, 2])),
    ('dec_width',      'v44k_width',  dict(seed=0, shape=[20000, 2])),
    ('dec_lr_width',   'v44k_lr_w',   dict(seed=0, shape=[20000, 2])),
    ('dec_ch0_lr',     'v44k_ch0_lr', dict(seed=0, shape=[20000, 2])),
    ('dec_nonorm',     'v44k_nonorm', dict(seed=0, shape=[20000, 2])),
    ('dec_c8_lr',      'v96k_c8',     dict(seed=0, shape=[30000, 8])),
    ('dec_zeros_mono', 'v44k',        dict(seed=0, shape=[1000], dist='zeros', dtype='float64')),
    # a single-channel table on an (n, 1) signal: NumPy sums that array pairwise (8192-sample chunks)
    ('dec_c1',         'v44k_c1',     dict(seed=0, shape=[20000, 1])),
    ('dec_c1_long',    'v44k_c1',     dict(seed=2, shape=[100001, 1])),
    ('dec_c1_tiny',    'v44k_c1',     dict(seed=3, shape=[5, 1])),
]


def oracle_class_kwargs(kw):
    k = dict(kw)
    norm = k.pop('normalizer', 'default')
    k['normalize'] = norm is not None
    return k


def taps_to_arrays(taps):
    """Flatten nested class taps for storage: rows (channel, segment, sign, idx)."""
    rows = []
    for c, segs in enumerate(taps):
        if segs is None:
            continue
        for s, (neg, pos) in enumerate(segs):
            rows += [(c, s, 0, i) for i in neg]
            rows += [(c, s, 1, i) for i in pos]
    return np.asarray(rows, np.int32).reshape(-1, 4)


def ref_taps_to_nested(vn):
    out = []
    for ch in vn.velvet_noise:
        if isinstance(ch, list) and len(ch) == 0:
            out.append(None)
            continue
        out.append([([int(i) for i in seg.negative_impulse_indexes],
                     [int(i) for i in seg.positive_impulse_indexes]) for seg in ch])
    return out


def main():
    dsp, dec, opt = load_reference()
    OUT.mkdir(parents=True, exist_ok=True)
    manifest = {'reference': 'ckonst/VNDecorrelate v1.1.0', 'numpy': np.__version__,
                'slice': SLICE, 'generators': {}, 'fn': {}, 'cls_convolve': {},
                'cls_decorrelate': {}, 'class_taps': {}, 'known_answers': {}, 'audio': {}, 'objective': {}, 'haas': {}}
    arrays = {}

    # ---- a2/a3/a4: generator ------------------------------------------------
    firs = {}
    for name, kw in GEN.items():
        ref = dec.generate_velvet_noise(**kw)
        mine = O.generate_velvet_noise(**kw)
        same(ref, mine, f'generate_velvet_noise[{name}]')
        firs[name] = ref
        offs, idx, w = O.fir_to_taps(ref)
        arrays[f'gen_{name}_offsets'] = offs
        arrays[f'gen_{name}_idx'] = idx
        arrays[f'gen_{name}_w'] = w
        manifest['generators'][name] = {'kwargs': {k: (list(v) if isinstance(v, tuple) else v)
                                                   for k, v in kw.items()},
                                        'fir_shape': list(ref.shape), 'fir_sha256': sha(ref)}
        print(f'gen  {name:14s} fir {ref.shape} taps/ch {np.diff(offs).tolist()}')

    # log-distribution identities the reference's own tests pin (tests/test_dsp.py:256-317)
    for strength, size in ((0.0, 30), (1.0, 30), (0.5, 128), (1.0, 64)):
        same(dsp.generate_log_distribution(strength, size), O.log_distribution(strength, size),
             f'log_distribution({strength},{size})')

    # ---- a1: function path ----------------------------------------------------
    for name, gname, spec in FN_CASES:
        x = make_input(spec)
        fir = firs[gname]
        ref = dec.convolve_velvet_noise(x, fir)
        mine = O.convolve_velvet_noise(x, fir)
        same(ref, mine, f'convolve_velvet_noise[{name}]')
        if x.dtype == np.float32 and x.size <= 200000:
            same(ref, O.convolve_taps_scalar(x, *O.fir_to_taps(fir)), f'scalar model[{name}]')
        manifest['fn'][name] = {'generator': gname, 'input': spec, 'out': pack_output(ref)}
        store_arrays(ref, name, arrays)
        print(f'fn   {name:16s} x {x.shape} {x.dtype} -> max|y| {manifest["fn"][name]["out"]["max_abs"]:.4f}')

    # f64 FIR (VelvetNoise.FIR) fed to the function path: weights stay float64
    vn = dec.VelvetNoise(**CLS['v44k'])
    x = make_input(dict(seed=0, shape=[10000, 2]))
    ref = dec.convolve_velvet_noise(x, vn.FIR)
    same(ref, O.convolve_velvet_noise(x, vn.FIR), 'convolve_velvet_noise[f64 FIR]')
    manifest['fn']['fn_f64_fir'] = {'class': 'v44k', 'input': dict(seed=0, shape=[10000, 2]),
                                    'out': pack_output(ref)}
    store_arrays(ref, 'fn_f64_fir', arrays)

    # batched (cfg4-shaped, 4 of the 1024 streams): per-stream independence
    xb = make_input(dict(seed=0, shape=[4, 48000, 2]))
    refb = np.stack([dec.convolve_velvet_noise(xb[b], firs['g48k_k30']) for b in range(4)])
    manifest['fn']['fn_cfg4_b4'] = {'generator': 'g48k_k30', 'input': dict(seed=0, shape=[4, 48000, 2]),
                                    'out': pack_output(refb),
                                    'per_stream_sha256': [sha(refb[b]) for b in range(4)]}
    arrays['fn_cfg4_b4_head'] = refb[:, :SLICE].copy()
    arrays['fn_cfg4_b4_tail'] = refb[:, -SLICE:].copy()

    # error behaviour (decorrelation.py:640-643, :656-657)
    try:
        dec.convolve_velvet_noise(np.zeros((10, 2), np.float32), firs['g96k_k64_c8'])
        raise SystemExit('expected ValueError')
    except ValueError:
        pass
    try:
        dec.convolve_velvet_noise(np.zeros(10, np.float32), firs['g44k_mono'])
        raise SystemExit('expected IndexError')
    except IndexError:
        pass

    # ---- a6/a7/a8: class path -------------------------------------------------
    for cname, kw in CLS.items():
        vn = dec.VelvetNoise(**kw)
        okw = {k: v for k, v in kw.items() if k not in ('width', 'mode', 'normalizer')}
        okw.setdefault('num_outs', 2)
        mine = O.generate_class_taps(**okw)
        reft = ref_taps_to_nested(vn)
        if reft != mine:
            raise SystemExit(f'ORACLE MISMATCH in class taps[{cname}]')
        env = tuple(vn.segment_envelope)
        same(vn.FIR, O.class_fir(mine, env, vn.fir_length_samples), f'FIR[{cname}]')
        arrays[f'taps_{cname}'] = taps_to_arrays(mine)
        manifest['class_taps'][cname] = {
            'kwargs': {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
            'fir_length_samples': vn.fir_length_samples, 'envelope': list(env),
            'fir_sha256': sha(vn.FIR), 'fir_shape': list(vn.FIR.shape)}
        print(f'taps {cname:12s} L={vn.fir_length_samples} rows={len(arrays[f"taps_{cname}"])}')

    for name, cname, spec in CLS_CONV_CASES:
        kw = CLS[cname]
        vn = dec.VelvetNoise(**kw)
        x = make_input(spec)
        ref = vn.convolve(x)
        okw = {k: v for k, v in kw.items() if k not in ('width', 'mode', 'normalizer')}
        okw.setdefault('num_outs', 2)
        taps = O.generate_class_taps(**okw)
        mine = O.class_convolve(x, taps, tuple(vn.segment_envelope), vn.num_outs)
        same(ref, mine, f'VelvetNoise.convolve[{name}]')
        manifest['cls_convolve'][name] = {'class': cname, 'input': spec, 'out': pack_output(ref)}
        store_arrays(ref, name, arrays)
        print(f'cls  {name:14s} x {x.shape} -> max|y| {manifest["cls_convolve"][name]["out"]["max_abs"]:.4f}')

    for name, cname, spec in CLS_DEC_CASES:
        kw = CLS[cname]
        vn = dec.VelvetNoise(**kw)
        x = make_input(spec)
        ref = vn.decorrelate(x.copy())
        mine = O.decorrelate(x.copy(), **oracle_class_kwargs(kw))
        same(ref, mine, f'VelvetNoise.decorrelate[{name}]')
        manifest['cls_decorrelate'][name] = {'class': cname, 'input': spec, 'out': pack_output(ref)}
        store_arrays(ref, name, arrays)
        print(f'dec  {name:14s} x {x.shape} -> max|y| {manifest["cls_decorrelate"][name]["out"]["max_abs"]:.4f}')

    # the reference's own equality tests (tests/test_decorrelation.py:71-93, :172-197)
    vn = dec.VelvetNoise(**CLS['v44k'])
    assert np.allclose(vn.FIR, firs['g44k_k30'], atol=1e-6)
    x = np.random.default_rng(0).random((10000, 2))
    assert np.allclose(dec.convolve_velvet_noise(x, firs['g44k_k30']), vn.convolve(x), atol=1e-6)

    # known answers held by the reference's tests
    vn55 = dec.VelvetNoise(sample_rate_hz=44100, duration_seconds=0.055, num_impulses=45)
    manifest['known_answers'] = {
        'density_30ms_30': dec.VelvetNoise(duration_seconds=0.03, num_impulses=30,
                                           sample_rate_hz=44100).density,
        'density_55ms_45': vn55.density, 'fir_shape_55ms': list(vn55.FIR.shape),
        'nonzeros_55ms': int(np.count_nonzero(vn55.FIR[:, 0])),
        'fn_fir_len_55ms': int(dec.generate_velvet_noise(duration_seconds=0.055, num_impulses=45).shape[0]),
    }
    try:
        dec.VelvetNoise(duration_seconds=0.03, num_impulses=700, sample_rate_hz=44100)
        raise SystemExit('expected ValueError (not sparse)')
    except ValueError:
        pass

    # ---- real audio: the reference's committed golden files ---------------------
    import scipy.io.wavfile as wavfile
    for stem in ('viola', 'vocal'):
        fs, audio = wavfile.read(REF / 'audio' / f'{stem}.wav')
        _, committed = wavfile.read(REF / 'audio' / f'{stem}_decorrelated.wav')
        chain = (dec.SignalChain(sample_rate_hz=fs)
                 .velvet_noise(**{k: v for k, v in CLS['v44k_20ms'].items() if k != 'sample_rate_hz'})
                 .haas_effect(delay_time_seconds=0.02, delayed_channel=1, mode='LR'))
        ref = chain(audio)
        same(ref, committed, f'reference chain vs committed {stem}_decorrelated.wav')
        mine = O.haas_delay_lr(O.decorrelate(audio, **oracle_class_kwargs(CLS['v44k_20ms'])),
                               sample_rate_hz=fs, delay_time_seconds=0.02, delayed_channel=1)
        same(mine, committed, f'oracle chain vs committed {stem}_decorrelated.wav')
        manifest['audio'][stem] = {'fs': int(fs), 'in_shape': list(audio.shape),
                                   'in_sha256': sha(audio), 'committed_shape': list(committed.shape),
                                   'committed_sha256': sha(committed),
                                   'reference_reproduces_committed': True,
                                   'oracle_reproduces_committed': True}
        print(f'wav  {stem}: reference and oracle both reproduce the committed output bit-for-bit')

    # 1 s viola excerpt (public-domain repo, LICENSE:1) pins real-audio behaviour on the GPU box
    fs, viola = wavfile.read(REF / 'audio' / 'viola.wav')
    start = 60000
    excerpt = np.ascontiguousarray(viola[start:start + fs])
    vn = dec.VelvetNoise(**CLS['v44k_20ms'])
    arrays['viola_excerpt_in'] = excerpt
    arrays['viola_excerpt_decorrelate'] = vn.decorrelate(excerpt.copy())
    arrays['viola_excerpt_convolve'] = vn.convolve(excerpt)
    arrays['viola_excerpt_fn'] = dec.convolve_velvet_noise(excerpt, firs['g44k_20ms'])
    same(arrays['viola_excerpt_decorrelate'],
         O.decorrelate(excerpt.copy(), **oracle_class_kwargs(CLS['v44k_20ms'])), 'viola excerpt')
    manifest['audio']['viola_excerpt'] = {'fs': int(fs), 'start_frame': start, 'class': 'v44k_20ms',
                                          'generator': 'g44k_20ms'}

    # ---- f3: the optimiser's candidate scan (optimization.py:46-117, :230-310) ---------------
    # the candidates of optimize_velvet_noise (:259-271) on a small kappa grid
    OBJ_KW = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0,
                  lambda_penalty=1e3)
    obj_inputs = {'viola_excerpt': excerpt,
                  'uniform_stereo': make_input(dict(seed=31, shape=[30000, 2])),
                  'uniform_mono': make_input(dict(seed=32, shape=[20000]))}
    for iname, sig in obj_inputs.items():
        fs_obj = 44100 if iname == 'viola_excerpt' else 48000
        kappas = np.linspace(0.0, 1.0, 9)
        cands = [dict(sample_rate_hz=fs_obj, duration_seconds=0.03, num_impulses=30,
                      log_distribution_strength=float(k), normalizer=None, filtered_channels=(0,),
                      mode='LR', seed=1) for k in kappas]
        scores, terms = [], []
        for kw in cands:
            vn = dec.VelvetNoise(**kw)
            ref_score = opt.symmetry_aware_objective(sig, vn, **OBJ_KW)
            out = vn.decorrelate(sig)
            mine = O.symmetry_aware_objective(out, **OBJ_KW)
            if ref_score != mine:
                raise SystemExit(f'ORACLE MISMATCH in objective[{iname}]: {ref_score!r} vs {mine!r}')
            _, th, wt = dsp.polar_coordinates(out[:, 0], out[:, 1], normalize=False)
            sp = opt.angular_variance(th, wt)
            ref_terms = (sp, opt.centroid(th, wt), opt.polar_skewness(th, wt, sp),
                         float(opt.left_right_correlation(out)), opt.max_angular_exceedance(th, OBJ_KW['angle_limit']))
            if ref_terms != tuple(float(v) for v in O.objective_terms(out, angle_limit=OBJ_KW['angle_limit'])):
                raise SystemExit(f'ORACLE MISMATCH in objective terms[{iname}]')
            scores.append(ref_score)
            terms.append(ref_terms)
        scan = opt.grid_scan(sig, [dec.VelvetNoise(**kw) for kw in cands], **OBJ_KW)
        assert np.array_equal(scan, np.array(scores))
        minima = opt.get_local_minima(scan, len(kappas))
        assert minima == O.local_minima(scan, len(kappas))
        arrays[f'obj_{iname}_scores'] = np.array(scores, np.float64)
        arrays[f'obj_{iname}_terms'] = np.array(terms, np.float64)
        manifest['objective'][iname] = {'sample_rate_hz': fs_obj, 'kappas': kappas.tolist(), 'kwargs': OBJ_KW,
                                        'local_minima': [int(i) for i in minima],
                                        'input': ('viola_excerpt_in' if iname == 'viola_excerpt' else
                                                  dict(seed=31, shape=[30000, 2]) if iname == 'uniform_stereo'
                                                  else dict(seed=32, shape=[20000]))}
        print(f'obj  {iname:14s} scores {np.array2string(np.array(scores), precision=4)} minima {minima}')
    # the whole optimiser on the excerpt, small grid (scipy's bounded Brent on the same objective)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        kappa_opt = opt.optimize_velvet_noise(input_signal=excerpt, sample_rate_hz=44100, duration_seconds=0.03,
                                              num_impulses=30, seed=1, grid_size=9)
    manifest['objective']['viola_excerpt']['optimize_velvet_noise_grid9'] = float(kappa_opt)
    print(f'obj  optimize_velvet_noise(grid 9) on the excerpt -> kappa {kappa_opt:.6f}')

    # ---- f4: HaasEffect, every mode (decorrelation.py:163-230) ---------------------------------
    HAAS = {
        'haas_lr_ch0': (dict(sample_rate_hz=48000, delay_time_seconds=0.01, delayed_channel=0, mode='LR'), [1500, 2]),
        'haas_lr_ch1_width': (dict(sample_rate_hz=44100, delay_time_seconds=0.02, delayed_channel=1, mode='LR',
                                   width=0.3), [1500, 2]),
        'haas_ms_mid': (dict(sample_rate_hz=48000, delay_time_seconds=0.005, delayed_channel=0, mode='MS'), [1500, 2]),
        'haas_ms_side_width': (dict(sample_rate_hz=48000, delay_time_seconds=0.0125, delayed_channel=1, mode='MS',
                                    width=0.7), [1500, 2]),
        'haas_mono_lr': (dict(sample_rate_hz=48000, delay_time_seconds=0.003, delayed_channel=1, mode='LR'), [1500]),
        'haas_mono_ms': (dict(sample_rate_hz=48000, delay_time_seconds=0.003, delayed_channel=1, mode='MS',
                              width=0.5), [1500]),
        'haas_zero_delay': (dict(sample_rate_hz=48000, delay_time_seconds=0.0, delayed_channel=0, mode='MS'), [100, 2]),
        'haas_longer_than_signal': (dict(sample_rate_hz=48000, delay_time_seconds=0.01, delayed_channel=0,
                                         mode='LR'), [100, 2]),
    }
    for name, (kw, shape) in HAAS.items():
        spec = dict(seed=50, shape=shape)
        x = make_input(spec)
        ref = dec.HaasEffect(**kw).decorrelate(x)
        mine = O.haas_effect(x, **kw)
        same(ref, mine, f'HaasEffect[{name}]')
        assert ref.dtype == np.float64
        arrays[f'{name}_out'] = ref
        manifest['haas'][name] = {'kwargs': kw, 'input': spec, 'out_shape': list(ref.shape), 'out_sha256': sha(ref)}
        print(f'haas {name:24s} x {x.shape} -> {ref.shape} {ref.dtype}')

    np.savez_compressed(OUT / 'golden.npz', **arrays)
    (OUT / 'manifest.json').write_text(json.dumps(manifest, indent=1, sort_keys=True) + '\n')
    size = (OUT / 'golden.npz').stat().st_size
    print(f'wrote {OUT}/golden.npz ({size/1e6:.2f} MB, {len(arrays)} arrays) and manifest.json')


if __name__ == '__main__':
    main()
