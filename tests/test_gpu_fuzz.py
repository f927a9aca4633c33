"""GPU tier: RANDOM tap tables - channel counts, spans, tap counts, weights (among them +-1, tiny ones, repeated magnitudes),
signal lengths and batches drawn per seed - through every kernel family the library can pick or be told to pick (automatic
choice, per-table pair-read form, per-table window form with 32- and 16-frame runs, generic kernels), against the NumPy
oracle: VND_MODE_EXACT bit for bit, VND_MODE_FAST within 1e-6 of the output peak.  The golden tables of the other tests
are the reference's; these are not velvet noise at all - any FIR table the API accepts must come out right."""
import numpy as np
import pytest

from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu

FORCE = 1 << 23
GENERIC = 1 << 25
WIN = {0: 1 << 5, 16: 2 << 5, 32: 3 << 5, 64: 4 << 5}


def span_bits(min_span, rounds):
    return (min_span << 20) | (rounds << 28)


# (channels, span of the offsets, most taps per channel): the shapes are laid out, what fills them is drawn
CASES = [(2, 700, 40), (8, 700, 24), (3, 64, 20), (2, 2500, 40), (4, 1500, 30), (6, 300, 16), (2, 8, 8), (1, 700, 30),
         (2, 3000, 60), (4, 8, 6),
         # halos that eight (four) channels cannot round up to a multiple of 16 ring entries inside 160 KB: the ring is cut and its last
         # entries live in a tail (WinGeom::tail) - random taps at every offset the tile's last lanes find there
         (8, 2719, 30), (8, 2650, 40), (4, 2719, 24), (16, 2719, 10),
         # 4k + 2 channels: k quads and one more that starts at channel C - 4 (two workgroups write the shared pair with the same bits)
         (6, 2719, 20), (10, 1500, 16), (14, 700, 10)]


@pytest.mark.parametrize('seed', range(len(CASES)))
def test_random_tables_through_every_kernel_family(seed, monkeypatch):
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    rng = np.random.default_rng(1000 + seed)
    ctx = _native.default_context()
    C, span, most = CASES[seed]
    fir = np.zeros((span, C), np.float32)
    for c in range(C):
        k = int(rng.integers(max(1, most // 2), min(most, span) + 1))
        idx = rng.choice(span, size=k, replace=False)
        w = rng.choice([1.0, -1.0, 0.5, -0.5, 0.85, -0.2, 1e-3, -3.0], size=k) * rng.choice([1.0, 1.0, rng.uniform(0.1, 1.0)], size=k)
        fir[idx, c] = w.astype(np.float32)
    if seed % 3 == 0:
        fir[0, :] = 0.75                                # a tap at offset 0 in every channel
    a = function_path_arrays(fir)
    table = _native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
    variants = [('automatic', -1), ('pair-read', FORCE | WIN[0] | span_bits(1, 3)), ('window 32', FORCE | WIN[32] | span_bits(1, 3)),
                ('window 16', FORCE | WIN[16] | span_bits(2, 1)), ('generic', GENERIC)]
    # the form that an environment switch selects (read live under VND_TUNING): stereo - 64-frame runs with the waves split over the
    # two channels
    env_of = {}
    if C == 2:
        variants.append(('window 64 split', FORCE | WIN[64] | span_bits(1, 3)))
        env_of['window 64 split'] = {'VND_WIN_SPLIT': '2', 'VND_SPEC_NT': '256'}
    try:
        # (the largest length is a multiple of 4 frames: streams of a batch then start 16-byte aligned, which the per-table
        #  kernels ask for - the odd lengths before it go through the generic kernels whatever is asked)
        for n in sorted({int(rng.integers(1, 200)), int(rng.integers(200, 9000)), 4 * int(rng.integers(2500, 17000))}):
            for batch in (1, 3):
                x = rng.uniform(-1, 1, (batch, n, C)).astype(np.float32)
                x[rng.integers(0, batch, 20), rng.integers(0, n, 20), rng.integers(0, C, 20)] = 0.0
                if seed % 2:
                    x[0, :min(n, 50)] *= np.float32(1e-30)      # products that underflow to denormals and to zero
                want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
                peak = float(np.max(np.abs(want))) or 1.0
                for name, variant in variants:
                    ctx.set_variant(variant)
                    for key in ('VND_WIN_SPLIT', 'VND_SPEC_NT'):
                        monkeypatch.delenv(key, raising=False)
                    for key, value in env_of.get(name, {}).items():
                        monkeypatch.setenv(key, value)
                    for mode in (d.MODE_EXACT, d.MODE_FAST):
                        got = table.convolve_host(x, mode)
                        where = f'seed {seed} C={C} span={span} taps={len(a.tap_index)} n={n} batch={batch} {name}: {table.describe(batch, n, C, mode)[:60]}'
                        assert got.shape == want.shape, where
                        if C % 2 == 0 and n >= 10000 and name != 'automatic':       # the family asked for is the family that ran
                            text = table.describe(batch, n, C, mode)
                            family = {'pair-read': 'conv_spec', 'generic': 'conv_'}.get(name, 'conv_spec')
                            assert text.startswith(family) and ('_window' in text) == name.startswith('window') and \
                                ('conv_spec' in text) == (name != 'generic'), where
                        if mode == d.MODE_EXACT:
                            assert np.array_equal(got, want), where
                        else:
                            err = float(np.max(np.abs(got.astype(np.float64) - want))) / peak
                            assert err <= 1e-6, f'{where}: {err:.2e}'
    finally:
        ctx.set_variant(-1)
        for key in ('VND_WIN_SPLIT', 'VND_SPEC_NT'):
            monkeypatch.delenv(key, raising=False)
        table.close()


# (sample rate, seconds of filter, impulses, segment envelope, log strength, width, mode, normalise, input: frames / channels)
CLASS_CASES = [
    (48000, 0.03, 30, None, 1.0, None, 'MS', True, (48000, 2)),
    (44100, 0.02, 12, (0.9, 0.5, 0.2), 0.5, 0.3, 'MS', True, (30011, 2)),
    (96000, 0.03, 64, (1.0,), 0.0, None, 'LR', True, (100000, 2)),
    (22050, 0.05, 40, (0.85, 0.55, 0.35, 0.2, 0.1), 1.0, 1.0, 'MS', False, (7001, 2)),
    (48000, 0.01, 5, (0.7, 0.7), 2.0, 0.0, 'LR', False, (500, 2)),
    (48000, 0.03, 30, None, 1.0, None, 'MS', True, (20000, 1)),          # a mono signal: duplicated to stereo first
    (32000, 0.04, 100, (1.0, 0.25), 0.3, 0.6, 'MS', True, (64000, 2)),
    (48000, 0.03, 30, None, 1.0, 0.5, 'LR', True, (200, 2)),             # shorter than the filter
]


@pytest.mark.parametrize('case', range(len(CLASS_CASES)))
def test_random_class_configurations_match_the_oracle(case):
    """VelvetNoise with fields away from the defaults (sample rate, filter length, impulse and segment counts, envelope,
    log-distribution strength, width, layout mode, normaliser off, seeds drawn per case) on random signals, against the
    oracle's restatement of the whole stage (tap generation included): bit for bit - the stage's default arithmetic is
    the reference's."""
    import vndecorrelate_amd.decorrelation as d
    rate, seconds, impulses, envelope, strength, width, mode, normalise, shape = CLASS_CASES[case]
    rng = np.random.default_rng(500 + case)
    for seed in (int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 31))):
        kw = dict(sample_rate_hz=rate, duration_seconds=seconds, num_impulses=impulses, log_distribution_strength=strength,
                  width=width, seed=seed)
        if envelope is not None:
            kw['segment_envelope'] = envelope
        vn = d.VelvetNoise(mode=d.LayoutMode[mode], **kw) if normalise else d.VelvetNoise(mode=d.LayoutMode[mode], normalizer=None, **kw)
        x = rng.uniform(-1, 1, shape if shape[1] == 2 else shape[:1]).astype(np.float32)
        want = O.decorrelate(x, mode=mode, normalize=normalise, **kw)
        got = vn.decorrelate(x)
        assert got.dtype == want.dtype and got.shape == want.shape, (case, seed)
        assert np.array_equal(got, want), f'case {case} seed {seed}: {float(np.max(np.abs(got - want))):.3e}'
        # convolve alone, and a second signal through the same (cached) table
        taps = O.generate_class_taps(sample_rate_hz=rate, duration_seconds=seconds, num_impulses=impulses,
                                     segment_envelope=envelope if envelope is not None else O.DEFAULT_ENVELOPE,
                                     log_distribution_strength=strength, seed=seed)
        x2 = rng.uniform(-1, 1, (shape[0] + 17, 2)).astype(np.float32)
        env = tuple(envelope) if envelope is not None else tuple(O.DEFAULT_ENVELOPE)
        assert np.array_equal(vn.convolve(x2), O.class_convolve(x2, taps, env, 2)), (case, seed)
