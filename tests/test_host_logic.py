"""CPU tier: host-side logic of the drop-in layer (no tap sum runs here):
generators against the reference's tables, the class shell, SignalChain glue,
tap-table builders and the epilogue helpers with the reference's known answers."""
import numpy as np
import pytest

import vndecorrelate_amd.decorrelation as vnd
from vndecorrelate_amd.taps import class_path_arrays, function_path_arrays
from vndecorrelate_amd.utils import dsp


def _kw(d):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}


# ---- a2/a3/a4: generator -------------------------------------------------------
def test_generate_velvet_noise_matches_reference(golden):
    for gname, meta in golden.manifest['generators'].items():
        fir = vnd.generate_velvet_noise(**_kw(meta['kwargs']))
        assert fir.dtype == np.float32 and np.array_equal(fir, golden.fir(gname)), gname


def test_log_distribution_identities():
    """tests/test_dsp.py:256-317 restated."""
    k, length = 30, 3000
    dist = dsp.generate_log_distribution(0, k)
    marks = np.cumsum(dist) - 1
    marks *= length / marks[-1]
    randoms = np.random.default_rng(1).uniform(0, 1, k + 1)
    a = dsp.apply_log_distribution(randoms, dist, marks, jitter=length / k)
    b = dsp.uniform_density(randoms, np.arange(k + 1), length / k)
    assert np.array_equal(a[:-1], b[:-1])
    c = dsp.apply_log_distribution(randoms, dist, marks, jitter=0.0)
    assert np.array_equal(c[:-1], marks[:-1].astype(np.int32))


# ---- a7/a8: VelvetNoise shell ----------------------------------------------------
def test_class_taps_match_reference(golden):
    for cname, meta in golden.manifest['class_taps'].items():
        kw = _kw(meta['kwargs'])
        vn = vnd.VelvetNoise(**kw)
        want = golden.class_taps(cname, vn.num_outs)
        got = [None if not len(seq) else
               [([int(i) for i in s.negative_impulse_indexes], [int(i) for i in s.positive_impulse_indexes])
                for s in seq] for seq in vn.velvet_noise]
        assert got == want, cname
        assert vn.fir_length_samples == meta['fir_length_samples']
        assert list(vn.FIR.shape) == meta['fir_shape']
        import hashlib
        assert hashlib.sha256(np.ascontiguousarray(vn.FIR).tobytes()).hexdigest() == meta['fir_sha256'], cname


def test_velvet_noise_properties(golden):
    """tests/test_decorrelation.py:96-115, :56-68, :161-169 restated."""
    assert vnd.VelvetNoise(duration_seconds=0.03, num_impulses=30, sample_rate_hz=44100).density == 1000
    vn = vnd.VelvetNoise(sample_rate_hz=44100, duration_seconds=0.055, num_impulses=45)
    assert 818.19 > vn.density > 818.18
    assert vn.FIR.shape == (2426, 2)
    assert len(vn.FIR[:, 0][vn.FIR[:, 0] != 0.0]) == 45
    assert vnd.generate_velvet_noise(duration_seconds=0.055, num_impulses=45).shape[0] == 2425
    with pytest.raises(ValueError):
        vnd.VelvetNoise(duration_seconds=0.03, num_impulses=700, sample_rate_hz=44100)
    a = vnd.VelvetNoise(sample_rate_hz=44100, seed=1)
    assert a._generate() == a._velvet_noise
    assert a._velvet_noise != vnd.VelvetNoise(sample_rate_hz=44100, seed=2)._velvet_noise
    assert a._velvet_noise != vnd.VelvetNoise(sample_rate_hz=44100, log_distribution_strength=0.0, seed=1)._velvet_noise
    first = a._velvet_noise
    _ = a.FIR
    assert a.velvet_noise is first                      # generated once
    a.num_impulses = 20                                 # ... and again when a key field changes
    assert a.velvet_noise is not first and a.velvet_noise.num_impluses == 20
    b = vnd.VelvetNoise(sample_rate_hz=44100, seed=1)
    kept = b.velvet_noise
    b.seed, b.log_distribution_strength = 99, 0.0       # these do NOT trigger regeneration upstream
    assert b.velvet_noise is kept
    vnd.VelvetNoise(sample_rate_hz=44100, segment_envelope=())
    assert vnd.VelvetNoise(sample_rate_hz=44100, segment_envelope=()).segment_envelope == (1.0,)
    with pytest.raises(IndexError):                     # RNG columns indexed by output channel (quirk 6)
        vnd.VelvetNoise(sample_rate_hz=44100, filtered_channels=(1,), seed=1)
    assert np.allclose(vnd.VelvetNoise(sample_rate_hz=44100, seed=1).FIR,
                       vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, seed=1), atol=1e-6)


# ---- tap-table builders ----------------------------------------------------------------
def test_function_path_arrays(golden):
    fir = golden.fir('g48k_k128_l')                     # duplicates already collapsed: 123 taps
    arr = function_path_arrays(fir)
    assert arr.tap_offsets.tolist() == [0, 123, 246] and arr.seg_offsets is None
    for c in range(2):
        sl = slice(arr.tap_offsets[c], arr.tap_offsets[c + 1])
        assert np.all(np.diff(arr.tap_index[sl]) > 0)
        assert np.array_equal(fir[arr.tap_index[sl], c], arr.tap_weight[sl])
    assert function_path_arrays(golden.fir('g96k_k64_c8'), 1).num_channels == 1


def test_class_path_arrays(golden):
    vn = vnd.VelvetNoise(**_kw(golden.manifest['class_taps']['v48k_k128_l']['kwargs']))
    arr = vn._tap_arrays()
    assert arr.tap_offsets.tolist() == [0, 128, 256]    # duplicates kept: 128 per channel
    assert arr.seg_offsets.tolist() == [0, 4, 8] and arr.apply_gain
    assert np.array_equal(arr.seg_gain[:4], np.float32([0.85, 0.55, 0.35, 0.2]))
    assert set(np.unique(arr.tap_weight)) == {-1.0, 1.0}
    assert arr.seg_end.tolist() == [32, 64, 96, 128, 160, 192, 224, 256]
    for s, seg in enumerate(vn.velvet_noise[0]):        # negatives first inside every segment
        sl = arr.tap_weight[32 * s:32 * (s + 1)]
        nneg = len(seg.negative_impulse_indexes)
        assert np.all(sl[:nneg] == -1.0) and np.all(sl[nneg:] == 1.0)
    one = vnd.VelvetNoise(sample_rate_hz=44100, seed=1, filtered_channels=(0,), mode='LR')._tap_arrays()
    assert one.chan_flags.tolist() == [0, 1] and one.tap_offsets.tolist() == [0, 30, 30]
    ident = vnd.VelvetNoise(sample_rate_hz=44100, seed=3, segment_envelope=())._tap_arrays()
    assert not ident.apply_gain and ident.seg_offsets.tolist() == [0, 1, 2]
    short = vnd.VelvetNoise(sample_rate_hz=44100, seed=5, num_impulses=15, duration_seconds=0.5,
                            segment_envelope=(1.0, 0.5, 0.25))
    short.segment_envelope = (0.5,)                      # shorter than the 3 generated segments
    with pytest.raises(IndexError):
        short._tap_arrays()
    short.segment_envelope = [1.0] * 1000                # a list is not the identity tuple: gains apply
    assert short._tap_arrays().apply_gain
    assert class_path_arrays([None, None], (1.0,), False).tap_offsets.tolist() == [0, 0, 0]


# ---- SignalChain glue ----------------------------------------------------------------------
def test_signal_chain_contract():
    with pytest.raises(TypeError):
        vnd.SignalChain(sample_rate_hz=44100, _decorrelators=[])
    chain = vnd.SignalChain(sample_rate_hz=44100)
    with pytest.raises(TypeError):
        chain.velvet_noise(sample_rate_hz=48000)
    chain.velvet_noise(sample_rate_hz=44100, seed=1).haas_effect(delay_time_seconds=0.02, delayed_channel=1)
    assert not chain._hot and callable(chain._decorrelators[0])
    chain._init_decorrelators()
    assert chain._hot and isinstance(chain._decorrelators[0], vnd.VelvetNoise)
    assert isinstance(chain._decorrelators[1], vnd.HaasEffect)
    eager = vnd.SignalChain(sample_rate_hz=44100, lazy=False).white_noise(seed=1)
    assert isinstance(eager._decorrelators[0], vnd.WhiteNoise)
    # partial semantics: positional extras come BEFORE the signal (reference quirk 7)
    seen = []
    out = vnd.SignalChain(sample_rate_hz=44100).stateless(lambda *a, **k: seen.append((a, k)) or a[-1], 'fir', gain=2)(
        np.arange(3))
    assert seen[0][0][0] == 'fir' and seen[0][1] == {'gain': 2} and np.array_equal(out, np.arange(3))


def test_numpy_only_stages():
    out = vnd.HaasEffect(sample_rate_hz=44100, delay_time_seconds=0.02)(np.ones(435))
    assert out.shape == (1317, 2) and out.dtype == np.float64 and out.sum() != 0.0
    x = np.random.default_rng(0).uniform(-1, 1, (500, 2)).astype(np.float32)
    lr = vnd.HaasEffect(sample_rate_hz=44100, delay_time_seconds=0.001, delayed_channel=1)(x)
    assert np.array_equal(lr[:500, 0], x[:, 0].astype(np.float64)) and np.array_equal(lr[44:544, 1], x[:, 1])
    wn = vnd.WhiteNoise(sample_rate_hz=44100, duration_seconds=0.03, seed=2)
    assert wn(np.random.default_rng(1).random(4350)).shape == (4350, 2) and wn.FIR.shape == (1323, 2)


# ---- a9: epilogue helpers, with the reference's known answers -------------------------
def test_dsp_helpers():
    x = np.column_stack((np.ones(100), np.zeros(100))).astype(np.float32)
    dsp.apply_stereo_width(x, 1.0)
    dsp.LR_to_MS(x)
    assert x[:, 0].sum() == 0.0 and x[:, 1].sum() != 0.0
    x = np.full((100, 2), 100).astype(np.float32)
    dsp.apply_stereo_width(x, 0.5)
    assert x.sum() == pytest.approx(100 * 100)
    a = np.array([[1, 2], [2, 4], [3, 6], [4, 8]]).astype(np.float32)
    dsp.LR_to_MS(a)
    assert np.array_equal(a, [[1.5, -0.5], [3, -1], [4.5, -1.5], [6, -2]])
    b = np.array([[1, 2], [2, 4], [3, 6], [4, 8]])
    dsp.MS_to_LR(b)
    assert np.array_equal(b, [[3, -1], [6, -2], [9, -3], [12, -4]])
    y = np.array([[1.0, 1.0], [1.0, 1.0]])
    dsp.rms_normalize(np.array([[0.707, 0.3535], [0.707, 0.3535]]), y, mode=dsp.NormalizeMode.STEREO)
    assert np.allclose(y, 0.55893258)
    y = np.array([[1.0, 1.0], [1.0, 1.0]])
    dsp.rms_normalize(np.array([[0.707, 0.3535], [0.707, 0.3535]]), y)
    assert np.allclose(y, [[0.707, 0.3535], [0.707, 0.3535]])
    f = np.array([1, 2, 3, 4], dtype=np.float32)
    assert dsp.to_float32(f) is f and dsp.to_float32(np.arange(4)).dtype == np.float32
    assert np.array_equal(dsp.mono_to_stereo(np.array([1, 2])), [[1, 1], [2, 2]])
    assert dsp.mono_to_stereo(np.array([])).shape == (0, 2)
    for bad in (np.array([]), np.zeros((3, 3))):
        with pytest.raises(ValueError):
            dsp.encode_signal_to_side_channel(np.zeros((3, 2)), bad)
    with pytest.raises(ValueError):
        dsp.mono_to_stereo(np.zeros((2, 2)))
    with pytest.raises(ValueError):
        dsp.stereo_to_mono(np.zeros(4))
    with pytest.raises(ValueError):
        dsp.check_equal_length(np.zeros((3, 2)), np.zeros((3, 8)), dim=1)
    assert dsp.LayoutMode.MS == 'MS' and str(dsp.LayoutMode.LR) == 'LR'
    p = np.array([[0.707, 0.3535], [0.707, 0.3535]])
    dsp.peak_normalize(p)
    assert np.allclose(p, 1.0)


def test_function_path_host_validation():
    """Shape errors are raised by the host layer before any device work."""
    fir8 = np.zeros((10, 8), np.float32)
    with pytest.raises(ValueError):
        vnd.convolve_velvet_noise(np.zeros((10, 2), np.float32), fir8)
    fir = np.zeros((10, 1), np.float32)
    fir[3] = 1.0
    with pytest.raises(IndexError):
        vnd.convolve_velvet_noise(np.zeros(10, np.float32), fir)
    assert vnd.convolve_velvet_noise(np.zeros(10, np.float32), np.zeros((10, 1), np.float32)).shape == (10,)
    assert vnd.convolve_velvet_noise(np.zeros((0, 2), np.float32), np.zeros((10, 2), np.float32)).shape == (0, 2)
    with pytest.raises(ValueError):
        vnd.convolve_velvet_noise_batched(np.zeros((4, 2), np.float32), fir)
    with pytest.raises(ValueError):
        vnd.set_default_mode(7)


# ---- filter banks: channel-wise concatenation of tap tables -------------------------
def test_concat_tap_arrays_against_the_oracle(golden):
    """A bank's table on the replicated signal == each member's table on the signal
    (checked with the C oracle: no GPU involved)."""
    from oracle import c_oracle
    from vndecorrelate_amd.taps import TapArrays, concat_tap_arrays
    x = np.random.default_rng(5).uniform(-1, 1, (3001, 2)).astype(np.float32)
    # function path
    firs = [vnd.generate_velvet_noise(duration_seconds=0.01, num_impulses=20 + f, sample_rate_hz=48000, seed=f)
            for f in range(3)]
    members = [function_path_arrays(f) for f in firs]
    bank = concat_tap_arrays(members)
    assert bank.num_channels == 6 and bank.seg_offsets is None
    y = c_oracle.convolve(np.tile(x, (1, 3)), bank.tap_offsets, bank.tap_index, bank.tap_weight)
    for f, m in enumerate(members):
        assert np.array_equal(y[:, 2 * f:2 * f + 2], c_oracle.convolve(x, m.tap_offsets, m.tap_index, m.tap_weight))
    assert TapArrays.from_bytes(bank.to_bytes()).tap_index.tolist() == bank.tap_index.tolist()
    # class path: gains, an identity envelope (gain skipped upstream), a pass-through channel
    vns = [vnd.VelvetNoise(sample_rate_hz=48000, seed=1),
           vnd.VelvetNoise(sample_rate_hz=48000, seed=2, segment_envelope=(1.0,)),
           vnd.VelvetNoise(sample_rate_hz=48000, seed=3, filtered_channels=(0,))]
    members = [v._tap_arrays() for v in vns]
    bank = concat_tap_arrays(members)
    assert bank.apply_gain and bank.chan_flags.tolist() == [0, 0, 0, 0, 0, 1]
    def oracle(sig, t):
        return c_oracle.convolve(sig, t.tap_offsets, t.tap_index, t.tap_weight, seg_off=t.seg_offsets,
                                 seg_end=t.seg_end, seg_gain=t.seg_gain, chan_flags=t.chan_flags,
                                 apply_gain=t.apply_gain)
    y = oracle(np.tile(x, (1, 3)), bank)
    for f, m in enumerate(members):
        want = oracle(x, m)
        assert np.array_equal(y[:, 2 * f:2 * f + 2], want), f
    with pytest.raises(ValueError):
        concat_tap_arrays([function_path_arrays(firs[0]), members[0]])
    with pytest.raises(ValueError):
        concat_tap_arrays([])


# ---- f4: HaasEffect (NumPy stage) against the reference's outputs -------------------
def test_haas_effect_matches_reference(golden):
    import hashlib
    from conftest import make_input
    from oracle import vnd_oracle as O
    for name, meta in golden.manifest['haas'].items():
        x = make_input(meta['input'])
        want = golden.arrays[f'{name}_out']
        got = vnd.HaasEffect(**_kw(meta['kwargs'])).decorrelate(x)
        assert got.dtype == np.float64 and list(got.shape) == meta['out_shape'], name
        assert np.array_equal(got, want), name
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == meta['out_sha256'], name
        assert np.array_equal(O.haas_effect(x, **meta['kwargs']), want), name
