"""GPU tier: the WINDOW form of the per-table kernel (vnd_win.hpp; VND_MODE_FAST's default for stereo tables) and
every per-table kernel at the pool shapes bench.py times, against the oracle.
Bar: <= 1e-6 of the output peak - the north star's tolerance, 128-tap tables included (2e-6 until round 4; measured: 7-8e-7
on the worst stream of the cfg3 pools) -; the exact kernels bit for bit."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu

TOL_PEAK = 1e-6
FORCE = 1 << 23            # specialise however little work there is
GENERIC = 1 << 25
WIN = {0: 1 << 5, 16: 2 << 5, 32: 3 << 5, 64: 4 << 5}      # variant bits 5-7: frames per lane (0: the pair-read kernel)


def span_bits(min_span, rounds):
    return (min_span << 20) | (rounds << 28)


@pytest.fixture(scope='module')
def env():
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    yield d, _native, ctx
    ctx.set_variant(-1)


def _table(native, ctx, fir):
    from vndecorrelate_amd.taps import function_path_arrays
    a = function_path_arrays(fir)
    return native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)


def _err(got, want):
    peak = float(np.max(np.abs(want))) or 1.0
    return float(np.max(np.abs(got.astype(np.float64) - want))) / peak


@pytest.mark.parametrize('gname', ['g48k_k30', 'g48k_k128_u', 'g44k_noenv'])
@pytest.mark.parametrize('M,nt', [(32, 192), (16, 128), (64, 64), (32, 128), (16, 64)])
def test_window_geometries_seams_and_tails(env, golden, monkeypatch, gname, M, nt):
    """Every length class around the tile (nt * M frames) and the ring, spans of one tile and more (ring refills,
    wraps of the ring's entry positions, units following each other in one workgroup), stream tails inside a lane's
    run, batches.  Checked against the NumPy oracle."""
    d, native, ctx = env
    fir = golden.fir(gname)
    table = _table(native, ctx, fir)
    monkeypatch.setenv('VND_SPEC_NT', str(nt))
    tol = TOL_PEAK
    rng = np.random.default_rng(5)
    T = nt * M
    lengths = [1, 2, 31, M - 1, M, M + 1, T - 1, T, T + 1, 2 * T + 3, 3 * T, 5 * T + 17, 9001, 12346, 40003]
    for n in sorted(set(lengths)):
        for batch in (1, 3):
            if batch > 1 and n % 2:
                continue                                  # odd stereo streams are 8-byte aligned: generic kernel
            x = rng.uniform(-1, 1, (batch, n, 2)).astype(np.float32)
            want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
            for min_span, rounds in ((1, 7), (2, 1)):
                ctx.set_variant(FORCE | WIN[M] | span_bits(min_span, rounds))
                text = table.describe(batch, n, 2, d.MODE_FAST)
                assert text.startswith('conv_spec_window') and f'frames_per_lane={M} ' in text and f'threads={nt}' in text, text
                got = table.convolve_host(x, d.MODE_FAST)
                assert _err(got, want) <= tol, f'{gname} M={M} nt={nt} n={n} batch={batch} spans=({min_span},{rounds}): {_err(got, want):.2e}'
    ctx.set_variant(-1)
    table.close()


def test_window_results_do_not_depend_on_the_geometry(env, golden, monkeypatch):
    """Every output is fl(E + O), E / O the fma chains over its even / odd taps in ascending offset - whatever the run
    length, the workgroup size, the span layout or the batch: bit-identical across all of them."""
    import torch
    d, native, ctx = env
    fir = golden.fir('g48k_k30')
    table = _table(native, ctx, fir)
    pool, n = 6, 100003 * 2
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    s = torch.cuda.current_stream().cuda_stream
    results = []
    for M, nt, spans in ((32, 256, (1, 2)), (32, 128, (2, 1)), (16, 128, (1, 5)), (64, 64, (3, 1)), (16, 64, (1, 1))):
        monkeypatch.setenv('VND_SPEC_NT', str(nt))
        ctx.set_variant(FORCE | WIN[M] | span_bits(*spans))
        assert f'frames_per_lane={M} ' in table.describe(pool, n, 2, d.MODE_FAST)
        y = torch.empty_like(x)
        table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=d.MODE_FAST, stream=s)
        torch.cuda.synchronize()
        results.append(y)
    for y in results[1:]:
        assert torch.equal(y, results[0])
    y1 = torch.empty((1, n, 2), dtype=torch.float32, device='cuda')
    table.convolve_device(x[4].data_ptr(), y1.data_ptr(), 1, n, 2, mode=d.MODE_FAST, stream=s)       # batch == loop
    torch.cuda.synchronize()
    assert torch.equal(y1[0], results[0][4])
    want = c_oracle.convolve(x[4].cpu().numpy(), *O.fir_to_taps(fir))
    assert _err(results[0][4].cpu().numpy(), want) <= TOL_PEAK
    ctx.set_variant(-1)
    table.close()


# ---- the per-table kernels at the pool shapes bench.py times (cfg3 twice, cfg4, cfg5), against the oracle ----
POOLS = {
    # name: (golden fir, pool, frames, channels, fast tolerance)
    'cfg3_uniform': ('g48k_k128_u', 24, 2880000, 2, 1e-6),        # 60 s stereo, 128 taps, kappa = 0
    'cfg3_log_123taps': ('g48k_k128_l', 24, 2880000, 2, 1e-6),    # kappa = 1: the function path keeps 123 of the 128 taps
    'cfg4_one_launch': ('g48k_k30', 1024, 48000, 2, 1e-6),        # 1024 streams of 1 s, device resident, ONE launch
    'cfg5_eight_channels': ('g96k_k64_c8', 16, 960000, 8, 1e-6),  # 96 kHz, 8 channels, 64 taps
    # (not bench shapes: the quad form where the quad IS the frame - non-temporal stores - and two octets per frame)
    'four_channels': ('g96k_k64_c8', 32, 480000, 4, 1e-6),
    'sixteen_channels': ('g96k_k64_c8', 8, 480000, 16, 1e-6),
    'six_channels': ('g96k_k64_c8', 32, 480000, 6, 1e-6),            # 4k + 2: two quads, channels 0-3 and 2-5
}


@pytest.mark.parametrize('name', list(POOLS))
def test_pools_at_bench_shapes_match_the_oracle(env, golden, name):
    """What bench.py's `secondary` and `cfg4_strong` legs launch: the automatic choice must be a per-table kernel in
    both modes; fast against the exact kernel on EVERY stream, three streams of both against the C oracle
    (exact: bit for bit; fast: the tolerance above)."""
    import torch
    d, native, ctx = env
    ctx.set_variant(-1)
    gname, pool, n, C, tol = POOLS[name]
    fir = golden.fir(gname)
    if C != fir.shape[1]:
        fir = np.ascontiguousarray(np.concatenate([fir, fir[:, ::-1]], axis=1)[:, :C])
    table = _table(native, ctx, fir)
    fast_text, exact_text = table.describe(pool, n, C, d.MODE_FAST), table.describe(pool, n, C, d.MODE_EXACT)
    assert fast_text.startswith('conv_spec'), fast_text
    assert exact_text.startswith('conv_spec_exact'), exact_text
    if C == 2:
        assert fast_text.startswith('conv_spec_window'), fast_text
    if C % 4 == 0 or C >= 6:
        assert fast_text.startswith('conv_spec_window') and exact_text.startswith('conv_spec_exact_window'), (fast_text, exact_text)
        pieces = 'pieces=channel-octets' if C % 8 == 0 else 'pieces=channel-quads'
        assert pieces in fast_text and pieces in exact_text, (fast_text, exact_text)
    x = torch.empty((pool, n, C), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y, ye = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, C, mode=d.MODE_FAST, stream=s)
    table.convolve_device(x.data_ptr(), ye.data_ptr(), pool, n, C, mode=d.MODE_EXACT, stream=s)
    torch.cuda.synchronize()
    peak = float(ye.abs().max())
    per_stream = (y - ye).abs().amax(dim=(1, 2))
    assert float(per_stream.max()) <= tol * peak, f'stream {int(per_stream.argmax())}: {float(per_stream.max()) / peak:.2e} of peak'
    offs, idx, w = O.fir_to_taps(fir)
    for b in (0, pool // 2, pool - 1):
        want = c_oracle.convolve(x[b].cpu().numpy(), offs, idx, w)
        assert np.array_equal(ye[b].cpu().numpy(), want), f'exact kernel, stream {b}'
        assert _err(y[b].cpu().numpy(), want) <= tol, f'fast kernel, stream {b}: {_err(y[b].cpu().numpy(), want):.2e}'
    table.close()


@pytest.mark.parametrize('gname', ['g48k_k128_u', 'g48k_k128_l'])
def test_fast_mode_margin_of_the_128_tap_tables_over_32_seeded_inputs(env, golden, gname):
    """The north star's 1e-6 on the tables where the margin is thinnest (128 taps: 6-8e-7 of peak on the bench pools): 32 seeded
    inputs - uniform, Gaussian, a decaying sine mix with noise, full-scale square-ish steps, eight of each - through the
    automatically chosen per-table kernel in one launch, EACH against the exact kernel (bit-identical to the C oracle: asserted on
    four of them) as a fraction of ITS OWN output peak.  The maxima are a distribution, not three seeds; it is written to
    gpurun_out/ when that directory exists (profiles/r05_k128_margin.json is a copy)."""
    import json
    import pathlib
    import torch
    d, native, ctx = env
    ctx.set_variant(-1)
    fir = golden.fir(gname)
    table = _table(native, ctx, fir)
    inputs, n = 32, 400000
    xs = np.empty((inputs, n, 2), np.float32)
    t = np.arange(n)
    for k in range(inputs):
        rng = np.random.default_rng(1000 + k)
        kind = k % 4
        if kind == 0:
            xs[k] = rng.uniform(-1, 1, (n, 2))
        elif kind == 1:
            xs[k] = np.clip(rng.standard_normal((n, 2)) * 0.3, -1, 1)
        elif kind == 2:
            f = rng.uniform(0.001, 0.2, 3)
            tone = sum(np.sin(t * fk) for fk in f)[:, None] * np.array([1.0, 0.7]) / 3.0
            xs[k] = (tone * np.exp(-t / (n / 3.0))[:, None] + 0.05 * rng.standard_normal((n, 2))).astype(np.float32)
        else:
            xs[k] = np.sign(rng.standard_normal((n, 2))) * rng.choice([0.25, 0.5, 1.0], (n, 1))
    text = table.describe(inputs, n, 2, d.MODE_FAST)
    assert text.startswith('conv_spec_window'), text
    x = torch.from_numpy(xs).cuda()
    y, ye = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    table.convolve_device(x.data_ptr(), y.data_ptr(), inputs, n, 2, mode=d.MODE_FAST, stream=s)
    table.convolve_device(x.data_ptr(), ye.data_ptr(), inputs, n, 2, mode=d.MODE_EXACT, stream=s)
    torch.cuda.synchronize()
    offs, idx, w = O.fir_to_taps(fir)
    for b in (0, 1, 2, 3):
        assert np.array_equal(ye[b].cpu().numpy(), c_oracle.convolve(xs[b], offs, idx, w, threads=8)), f'exact kernel, input {b}'
    rel = ((y - ye).abs().amax(dim=(1, 2)) / ye.abs().amax(dim=(1, 2))).cpu().numpy().astype(np.float64)
    out = pathlib.Path(__file__).resolve().parents[1] / 'gpurun_out'
    if out.is_dir():
        (out / f'r5_k128_margin_{gname}.json').write_text(json.dumps({
            'table': gname, 'inputs': inputs, 'frames': n, 'launch': text, 'kinds': ['uniform', 'gaussian', 'decaying tones + noise', 'steps'],
            'max_error_of_own_peak_per_input': [float(f'{v:.4g}') for v in rel],
            'max': float(rel.max()), 'median': float(np.median(rel)), 'min': float(rel.min())}, indent=1))
    worst = int(rel.argmax())
    assert rel[worst] <= TOL_PEAK, f'input {worst} (kind {worst % 4}): {rel[worst]:.3e} of its peak'
    table.close()


# ---- VND_MODE_EXACT in the window form: the reference's association, bit for bit ----
@pytest.mark.parametrize('gname', ['g48k_k30', 'g48k_k128_u', 'g48k_k128_l', 'g44k_noenv'])
@pytest.mark.parametrize('M,nt', [(32, 192), (16, 128), (32, 256)])
def test_exact_mode_window_is_bit_identical(env, golden, monkeypatch, gname, M, nt):
    """One accumulator per output, taps in table order, separately rounded products and sums, accumulators opened from
    zero: the window form in exact mode must equal the oracle bit for bit - through ring refills and wraps, stream tails
    inside a run and batches."""
    d, native, ctx = env
    fir = golden.fir(gname)
    table = _table(native, ctx, fir)
    monkeypatch.setenv('VND_SPEC_NT', str(nt))
    rng = np.random.default_rng(23)
    T = nt * M
    for n in sorted({1, 2, 31, M + 1, T - 1, T, 2 * T + 3, 9001, 12346, 40003}):
        for batch in (1, 3):
            if batch > 1 and n % 2:
                continue
            x = rng.uniform(-1, 1, (batch, n, 2)).astype(np.float32)
            x[rng.integers(0, batch, 50), rng.integers(0, n, 50), rng.integers(0, 2, 50)] = 0.0
            want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
            for min_span, rounds in ((1, 7), (2, 1)):
                ctx.set_variant(FORCE | WIN[M] | span_bits(min_span, rounds))
                text = table.describe(batch, n, 2, d.MODE_EXACT)
                assert text.startswith('conv_spec_exact_window') and f'frames_per_lane={M} ' in text, text
                got = table.convolve_host(x, d.MODE_EXACT)
                assert np.array_equal(got, want), f'{gname} M={M} nt={nt} n={n} batch={batch} spans=({min_span},{rounds})'
    ctx.set_variant(-1)
    table.close()


@pytest.mark.parametrize('name', sorted(__import__('json').loads((__import__('pathlib').Path(__file__).parent / 'golden' / 'manifest.json').read_text())['cls_convolve']))
def test_exact_mode_window_class_path(env, golden, name):
    """VelvetNoise.convolve's association (segments of -/+ unit taps, one multiply per segment, segments summed;
    decorrelation.py:402-414) through the window form: the reference's sha256."""
    from conftest import make_input
    d, native, ctx = env
    meta = golden.manifest['cls_convolve'][name]
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps'][meta['class']]['kwargs'].items()}
    vn = d.VelvetNoise(**kw)
    x = make_input(meta['input'])
    ctx.set_variant(FORCE | WIN[32] | span_bits(1, 3))
    try:
        y = vn.convolve(x)
        in_scope = len(kw.get('filtered_channels', (0, 1))) == vn.num_outs == 2 and vn._device_table().max_index < 3000
        launch = vn._device_table().describe(1, len(x), vn.num_outs, d.MODE_EXACT)
        if in_scope and (x.ndim == 2 and x.shape[1] == 2):
            assert launch.startswith('conv_spec_exact_window'), (name, launch)
    finally:
        ctx.set_variant(-1)
    golden.expect(name, y, exact=x.dtype == np.float32, rtol_peak=TOL_PEAK)


@pytest.mark.parametrize('C', [4, 6, 8])
@pytest.mark.parametrize('M,nt', [(32, 128), (16, 128)])
def test_window_form_on_wider_signals(env, golden, monkeypatch, C, M, nt):
    """More than two interleaved channels: a workgroup takes ONE channel pair of a span (8 bytes of every frame, its own
    pair's tap function).  Every pair has its own taps; lengths around the tile, stream tails inside a run, batches,
    spans of one tile - fast within tolerance, exact bit for bit, against the NumPy oracle."""
    d, native, ctx = env
    fir = np.ascontiguousarray(golden.fir('g96k_k64_c8')[:, :C])
    table = _table(native, ctx, fir)
    monkeypatch.setenv('VND_SPEC_NT', str(nt))
    monkeypatch.setenv('VND_WIN_QUAD', '0')           # (4 and 8 channels would take the quad form below)
    rng = np.random.default_rng(C * 100 + M)
    T = nt * M
    for n in sorted({1, 3, M + 1, T - 1, T, T + 1, 2 * T + 3, 5 * T + 17, 40003}):
        for batch in (1, 3):
            x = rng.uniform(-1, 1, (batch, n, C)).astype(np.float32)
            want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
            for min_span, rounds in ((1, 7), (2, 1)):
                ctx.set_variant(FORCE | WIN[M] | span_bits(min_span, rounds))
                for mode, name in ((d.MODE_FAST, 'conv_spec_window'), (d.MODE_EXACT, 'conv_spec_exact_window')):
                    text = table.describe(batch, n, C, mode)
                    assert text.startswith(name) and f'frames_per_lane={M} ' in text and f'threads={nt}' in text, text
                    assert 'channel-quads' not in text, text
                    got = table.convolve_host(x, mode)
                    where = f'C={C} M={M} n={n} batch={batch} spans=({min_span},{rounds})'
                    if mode == d.MODE_EXACT:
                        assert np.array_equal(got, want), where
                    else:
                        assert _err(got, want) <= TOL_PEAK, f'{where}: {_err(got, want):.2e}'
    ctx.set_variant(-1)
    table.close()


@pytest.mark.parametrize('C,M,nt,Q', [(8, 32, 512, 2), (16, 32, 512, 2), (8, 16, 512, 2), (4, 32, 256, 1), (12, 32, 256, 1), (4, 16, 256, 1), (8, 32, 256, 1), (16, 16, 256, 1),
                                      (6, 32, 256, 1), (10, 32, 256, 1), (6, 16, 256, 1)])      # 4k + 2 channels: k quads and one more from channel C - 4 (overlapping in one pair)
def test_window_form_on_channel_quads(env, golden, monkeypatch, C, M, nt, Q):
    """Signals of 4k interleaved channels: a workgroup takes a channel QUAD of a span - 16 bytes of every frame - or (Q = 2, signals
    of 8k channels) an OCTET: two neighbouring quads, 32 bytes of every frame, with 8 channels whole frames.  Its waves are split
    over its 4Q CHANNELS (vw_span_qc: one channel's accumulators per lane, 32-frame runs), outputs exchanged between the partner
    waves through the tile's dead ring entries.  Every channel has its own taps; lengths around the tile (nt / 4Q entries of M
    frames), stream tails inside a run, batches, spans of one tile and span seams - fast within tolerance, exact bit for bit,
    against the NumPy oracle."""
    d, native, ctx = env
    wide = golden.fir('g96k_k64_c8')
    fir = np.ascontiguousarray(np.concatenate([wide, wide[:, ::-1]], axis=1)[:, :C])
    table = _table(native, ctx, fir)
    monkeypatch.setenv('VND_SPEC_NT', str(nt))
    monkeypatch.setenv('VND_WIN_OCTET', '1' if Q == 2 else '0')
    rng = np.random.default_rng(C * 1000 + M + Q + 7)
    T = (nt // (4 * Q)) * M
    pieces = 'pieces=channel-octets' if Q == 2 else 'pieces=channel-quads'
    for n in sorted({1, 3, M + 1, T - 1, T, T + 1, 2 * T + 3, 5 * T + 17, 40003}):
        for batch in (1, 3):
            x = rng.uniform(-1, 1, (batch, n, C)).astype(np.float32)
            want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
            for min_span, rounds in ((1, 7), (2, 1)):
                ctx.set_variant(FORCE | WIN[M] | span_bits(min_span, rounds))
                for mode, name in ((d.MODE_FAST, 'conv_spec_window'), (d.MODE_EXACT, 'conv_spec_exact_window')):
                    text = table.describe(batch, n, C, mode)
                    assert text.startswith(name) and f'frames_per_lane={M} ' in text and f'tile={T} ' in text, text
                    assert f'threads={nt}' in text and pieces in text and 'waves=split-by-channel' in text, text
                    got = table.convolve_host(x, mode)
                    where = f'C={C} M={M} n={n} batch={batch} spans=({min_span},{rounds})'
                    if mode == d.MODE_EXACT:
                        assert np.array_equal(got, want), where
                    else:
                        assert _err(got, want) <= TOL_PEAK, f'{where}: {_err(got, want):.2e}'
    ctx.set_variant(-1)
    table.close()


@pytest.mark.parametrize('gname,M,nt', [('g48k_k128_u', 32, 256), ('g48k_k30', 16, 256), ('g48k_k128_l', 32, 512), ('g44k_noenv', 32, 384),
                                        ('g48k_k30', 64, 256), ('g48k_k128_u', 64, 256), ('g44k_k30', 64, 128)])
def test_window_form_with_the_waves_split_over_the_channels(env, golden, monkeypatch, gname, M, nt):
    """VND_WIN_SPLIT=2: half a workgroup's waves compute channel 0 of the tile's nt / 2 entries, the others channel 1 (one
    channel's accumulators per lane: three waves per SIMD), outputs exchanged through the tile's dead ring entries (vw_span_s).
    Lengths around the tile, tails, batches, one-tile spans and seams against the NumPy oracle - exact bit for bit, fast within
    tolerance.  A fast-mode build of this form that spills is rejected like any window build: the launch takes the plain form and
    is still right.  Off by default (profiles/r03_split_waves.txt)."""
    d, native, ctx = env
    fir = golden.fir(gname)
    table = _table(native, ctx, fir)
    monkeypatch.setenv('VND_SPEC_NT', str(nt))
    monkeypatch.setenv('VND_WIN_SPLIT', '2')
    tol = TOL_PEAK
    rng = np.random.default_rng(M + nt)
    T = (nt // 2) * M
    split_seen = 0
    for n in sorted({1, 3, M + 1, T - 1, T, T + 1, 2 * T + 3, 5 * T + 17, 40003}):
        for batch in (1, 3):
            if batch > 1 and n % 2:
                continue
            x = rng.uniform(-1, 1, (batch, n, 2)).astype(np.float32)
            want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
            for min_span, rounds in ((1, 7), (2, 1)):
                ctx.set_variant(FORCE | WIN[M] | span_bits(min_span, rounds))
                for mode, name in ((d.MODE_FAST, 'conv_spec_window'), (d.MODE_EXACT, 'conv_spec_exact_window')):
                    text = table.describe(batch, n, 2, mode)
                    if 'waves=split-by-channel' in text:
                        assert text.startswith(name) and f'frames_per_lane={M} ' in text, text
                        assert f'tile={T} ' in text and f'threads={nt}' in text, text
                        split_seen += 1
                    else:
                        # (only fast-mode builds of this form have been seen to spill; with 64-frame runs the plain window form
                        #  spills as well and the launch ends at the pair-read kernel)
                        assert mode == d.MODE_FAST and text.startswith('conv_spec'), text
                    got = table.convolve_host(x, mode)
                    where = f'{gname} M={M} n={n} batch={batch} spans=({min_span},{rounds})'
                    if mode == d.MODE_EXACT:
                        assert np.array_equal(got, want), where
                    else:
                        assert _err(got, want) <= tol, f'{where}: {_err(got, want):.2e}'
    assert split_seen > 0
    ctx.set_variant(-1)
    monkeypatch.delenv('VND_WIN_SPLIT')
    monkeypatch.delenv('VND_SPEC_NT')
    # by default: 64-frame runs in this form (two waves per SIMD) - the exact mode of a function-path table, and the fast mode
    # wherever that build does not spill (a table whose build does spill keeps the plain 32-frame form)
    fast = table.describe(24, 2880000, 2, d.MODE_FAST)
    assert ('waves=split-by-channel' in fast and 'frames_per_lane=64 ' in fast) or ('frames_per_lane=32 ' in fast and 'split' not in fast), fast
    text = table.describe(24, 2880000, 2, d.MODE_EXACT)
    assert 'waves=split-by-channel' in text and 'frames_per_lane=64 ' in text and 'threads=256' in text, text
    table.close()


def test_stereo_tables_take_the_window_form_by_default_in_both_modes(env, golden):
    """The automatic choice for a stereo table with enough work: the window form, fast and exact, function path and class
    path; tables of 4k channels the window form on channel quads, of 8k channels on octets; a mono input fanned out the plain form
    with one read stream for both output channels (fast; exact: function-path tables) or 64-frame split runs (exact, class-path
    tables); tables of 4k + 2 channels ride quads too."""
    d, native, ctx = env
    ctx.set_variant(-1)
    dense, sparse = _table(native, ctx, golden.fir('g48k_k128_u')), _table(native, ctx, golden.fir('g48k_k30'))
    wide = _table(native, ctx, golden.fir('g96k_k64_c8'))
    six = _table(native, ctx, np.ascontiguousarray(golden.fir('g96k_k64_c8')[:, :6]))
    four = _table(native, ctx, np.ascontiguousarray(golden.fir('g96k_k64_c8')[:, :4]))
    cls = d.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()
    for table, shape in ((dense, (24, 2880000, 2)), (sparse, (128, 480000, 2)), (cls, (128, 480000, 2))):
        exact = table.describe(*shape, d.MODE_EXACT)
        assert exact.startswith('conv_spec_exact_window'), exact
        # (64-frame runs with the waves split over the channels - function-path tables, and since round 6 the class path's segments
        #  too, with the fast mode's late refill; a build that spills would fall back to the plain form)
        if table is not cls:
            assert 'waves=split-by-channel' in exact and 'frames_per_lane=64 ' in exact, exact
        else:
            assert ('frames_per_lane=64 ' in exact and 'waves=split-by-channel' in exact) or ('frames_per_lane=32 ' in exact and 'split' not in exact), exact
        fast = table.describe(*shape, d.MODE_FAST)
        assert fast.startswith('conv_spec_window'), fast
        # (fast mode: 64-frame runs in the split form - the function-path tables build without spilling; a table whose build
        #  spills keeps the plain form with 32-frame runs)
        if table is not cls:
            assert 'frames_per_lane=64 ' in fast and 'waves=split-by-channel' in fast, fast
        else:
            assert ('frames_per_lane=64 ' in fast and 'waves=split-by-channel' in fast) or ('frames_per_lane=32 ' in fast and 'split' not in fast), fast
    for mode in (d.MODE_FAST, d.MODE_EXACT):
        text = sparse.describe(128, 480000, 1, mode)
        assert text.startswith('conv_spec') and 'window' in text, text
        assert 'frames_per_lane=32 ' in text and 'split' not in text, text          # (one merged read stream, both modes: function-path table)
        if mode == d.MODE_EXACT:
            text = cls.describe(128, 480000, 1, mode)                              # (class-path table, exact: the split form, input staged twice)
            assert text.startswith('conv_spec_exact') and (('frames_per_lane=64 ' in text and 'waves=split-by-channel' in text) or 'window' not in text), text
        text = wide.describe(16, 960000, 8, mode)
        assert text.startswith('conv_spec') and 'window' in text and 'pieces=channel-octets waves=split-by-channel' in text and 'frames_per_lane=32 ' in text, text
        text = four.describe(16, 960000, 4, mode)
        assert text.startswith('conv_spec') and 'window' in text and 'pieces=channel-quads waves=split-by-channel' in text and 'frames_per_lane=32 ' in text, text
        text = six.describe(16, 960000, 6, mode)                   # 4k + 2 channels: quads too, the last one overlapping (until round 5: the pair-read form)
        assert text.startswith('conv_spec') and 'window' in text and 'pieces=channel-quads' in text, text
    dense.close(); sparse.close(); wide.close(); six.close(); four.close()


def test_a_window_build_that_spills_is_rejected_and_the_launch_still_right(env, golden, monkeypatch):
    """The window form lives on its registers (about 240 of the 256 a lane has at two waves per SIMD with 32-frame runs): a
    build that spills is refused - here the one geometry known to, 32-frame runs in one-wave workgroups - and the launch
    takes the next geometry or the pair-read form; never a wrong result, never a CPU path."""
    d, native, ctx = env
    fir = golden.fir('g48k_k30')
    table = _table(native, ctx, fir)
    monkeypatch.setenv('VND_SPEC_NT', '64')
    x = np.random.default_rng(8).uniform(-1, 1, (3, 40000, 2)).astype(np.float32)
    want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(3)])
    ctx.set_variant(FORCE | WIN[32])
    text = table.describe(3, 40000, 2, d.MODE_FAST)
    assert text.startswith('conv_spec'), text                      # (whichever form was built: a per-table kernel)
    assert _err(table.convolve_host(x, d.MODE_FAST), want) <= TOL_PEAK
    ctx.set_variant(-1)
    table.close()


def test_prepare_builds_the_kernel_a_small_launch_will_take(env, golden, tmp_path, monkeypatch):
    """vnd_prepare_launch: a launch too small to stall for a hipRTC build (a rank's shard, one file) keeps the generic
    kernel until the host prepares its shape - once, off the hot path; then launches of that shape take the per-table
    kernel.  Results: bit-identical in exact mode, within tolerance in fast mode, prepared or not."""
    import torch
    d, native, ctx = env
    monkeypatch.setenv('VND_SPEC_CACHE_DIR', str(tmp_path / 'cache'))
    fir = d.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=4242)
    table = _table(native, ctx, fir)
    pool, n = 128, 48000                                   # the N = 8 shard of cfg4: 6 M frames
    ctx.set_variant(-1)
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    for mode, generic in ((d.MODE_FAST, 'conv_fast'), (d.MODE_EXACT, 'conv_ordered')):
        assert table.describe(pool, n, 2, mode).startswith(generic)
        table.convolve_device(x.data_ptr(), y0.data_ptr(), pool, n, 2, mode=mode, stream=s)
        table.prepare(pool, n, 2, mode)
        assert table.describe(pool, n, 2, mode).startswith('conv_spec'), table.describe(pool, n, 2, mode)
        table.convolve_device(x.data_ptr(), y1.data_ptr(), pool, n, 2, mode=mode, stream=s)
        torch.cuda.synchronize()
        if mode == d.MODE_EXACT:
            assert torch.equal(y0, y1)
        else:
            assert float((y0 - y1).abs().max()) <= TOL_PEAK * float(y0.abs().max())
    table.prepare(1, 100, 2, d.MODE_FAST)                  # nothing applies to so little work: not an error
    with pytest.raises(ValueError):
        table.prepare(4, 1000, 3, d.MODE_FAST)             # a shape the table cannot take
    table.close()


@pytest.mark.parametrize('M', [32, 16])
def test_mono_fan_out_through_the_window_form_when_forced(env, golden, M):
    """A mono input through a stereo table keeps the pair-read form by default (its two channels share the reads of the one
    plane); the window form's one-plane variant (VW_BC: one ring plane, a pass per channel, the transposition in two
    rounds) stays correct - forced here: fast mode within tolerance, exact mode bit for bit, seams and tails included."""
    d, native, ctx = env
    fir = golden.fir('g48k_k30')
    table = _table(native, ctx, fir)
    rng = np.random.default_rng(77)
    for n, batch in ((40003, 1), (12346, 3), (2 * 128 * M + 6, 2)):          # (batches of a mono input: even lengths keep the streams 8-byte aligned)
        x = rng.uniform(-1, 1, (batch, n, 1)).astype(np.float32)
        want = np.stack([O.convolve_velvet_noise(np.repeat(x[b], 2, axis=1), fir) for b in range(batch)])
        for spans in ((1, 7), (2, 1)):
            ctx.set_variant(FORCE | WIN[M] | span_bits(*spans))
            assert table.describe(batch, n, 1, d.MODE_FAST).startswith('conv_spec_window'), table.describe(batch, n, 1, d.MODE_FAST)
            assert _err(table.convolve_host(x, d.MODE_FAST), want) <= TOL_PEAK
            assert table.describe(batch, n, 1, d.MODE_EXACT).startswith('conv_spec_exact_window')
            assert np.array_equal(table.convolve_host(x, d.MODE_EXACT), want), (n, batch, spans)
    ctx.set_variant(-1)
    table.close()


@pytest.mark.parametrize('streams,n', [(128, 48000), (64, 96000), (256, 20000), (128, 40004)])
def test_small_one_round_launches_are_cut_into_cu_chunks(env, golden, streams, n):
    """A launch whose tiles do not fill one round of workgroups evenly (cfg4's N = 8 shard: 3 tiles of 8192 frames per CU) gives
    every CU a CHUNK of consecutive tiles, split between its two co-resident workgroups - the longer piece to the one dispatched
    first, the other starting a few microseconds later (make_spec_plan, vnd_win_kernel.inc).  Every stream of such launches
    against the C oracle: exact mode bit for bit (function- and class-path tables), fast mode within tolerance; the pieces'
    seams fall inside streams, the last chunk of a stream is ragged, and the plan says what it did."""
    d, native, ctx = env
    from vndecorrelate_amd.taps import function_path_arrays
    ctx.set_variant(-1)
    fir = golden.fir('g48k_k30')
    a = function_path_arrays(fir)
    table = native.TapTable.create(ctx, a.tap_offsets, a.tap_index, a.tap_weight)
    cls = d.VelvetNoise(sample_rate_hz=48000, seed=1)
    cls_table = cls._device_table()
    rng = np.random.default_rng(streams + n)
    x = rng.uniform(-1, 1, (streams, n, 2)).astype(np.float32)
    want = c_oracle.convolve(x, a.tap_offsets, a.tap_index, a.tap_weight, threads=8)
    for mode in (d.MODE_FAST, d.MODE_EXACT):
        table.prepare(streams, n, 2, mode)
        text = table.describe(streams, n, 2, mode)
        assert text.startswith('conv_spec') and '_window' in text and 'a chunk of 3 tiles per CU as 2 + 1' in text, text
        got = table.convolve_host(x, mode)
        if mode == d.MODE_EXACT:
            assert np.array_equal(got, want), f'{streams} x {n}'
        else:
            worst = max(_err(got[b], want[b]) for b in range(streams))
            assert worst <= TOL_PEAK, f'{streams} x {n}: {worst:.2e}'
    # the class path's table (segments, +-1 weights: the plain 32-frame form in exact mode)
    taps = O.generate_class_taps(sample_rate_hz=48000, seed=1)
    cls_table.prepare(streams, n, 2, d.MODE_EXACT)
    text = cls_table.describe(streams, n, 2, d.MODE_EXACT)
    assert 'a chunk of 3 tiles per CU as 2 + 1' in text, text
    got = cls_table.convolve_host(x, d.MODE_EXACT)
    for b in sorted({0, 1, streams // 2, streams - 1}):
        assert np.array_equal(got[b], O.class_convolve(x[b], taps, tuple(O.DEFAULT_ENVELOPE), 2)), (streams, n, b)
    # ... and a mono input fanned out (one ring plane), forced into the window form
    xm = np.ascontiguousarray(x[:, :, :1])
    ctx.set_variant(WIN[32])
    try:
        table.prepare(streams, n, 1, d.MODE_EXACT)
        text = table.describe(streams, n, 1, d.MODE_EXACT)
        if '_window' in text:
            assert 'a chunk of' in text, text
        got = table.convolve_host(xm, d.MODE_EXACT)
        wantm = c_oracle.convolve(np.ascontiguousarray(np.repeat(xm, 2, axis=2)), a.tap_offsets, a.tap_index, a.tap_weight, threads=8)
        assert np.array_equal(got, wantm)
    finally:
        ctx.set_variant(-1)
    table.close()


def test_round6_forms_against_the_forms_they_replaced(env, golden, monkeypatch):
    """Round 6's three default changes, each beside its switch (tuning variables, read live):
    * VND_MODE_FAST in the reference's class-path association - adds inside a run of equal |w|, the gain ratio once per run
      (decorrelation.py:402-414) - against one FMA per tap (VND_WIN_ADDS=0): both within 1e-6 of peak of the oracle on every
      stream, and within 6e-7 of each other (two summation orders of the same products);
    * VND_MODE_EXACT of a mono input through a function-path table with ONE merged read stream and shared products against a pass
      per channel in the split form (VND_WIN_EXACT_MERGED=0);
    * VND_MODE_EXACT of a class-path table in the split 64-frame form against the plain 32-frame form (VND_WIN_SPLIT_CLASS=0):
    the exact ones bit for bit the oracle either way."""
    d, native, ctx = env
    ctx.set_variant(FORCE)
    try:
        rng = np.random.default_rng(66)
        for gname, shape in (('g48k_k30', (6, 70001 * 2, 2)), ('g48k_k128_u', (3, 90000, 2)), ('g96k_k64_c8', (2, 50000, 8))):
            fir = golden.fir(gname)
            table = _table(native, ctx, fir)
            x = rng.uniform(-1, 1, shape).astype(np.float32)
            want = c_oracle.convolve(x, *O.fir_to_taps(fir), threads=4)
            got = {}
            for adds in ('1', '0'):
                monkeypatch.setenv('VND_WIN_ADDS', adds)
                text = table.describe(*shape, d.MODE_FAST)
                assert text.startswith('conv_spec_window') and text.endswith('taps=adds-per-segment') == (adds == '1'), text
                got[adds] = table.convolve_host(x, d.MODE_FAST)
                for b in range(shape[0]):
                    assert _err(got[adds][b], want[b]) <= TOL_PEAK, (gname, adds, b)
            assert 0.0 < _err(got['1'], got['0'].astype(np.float64)) <= 6e-7, gname
            monkeypatch.delenv('VND_WIN_ADDS')
            table.close()
        # a mono input through the function-path table, exact
        fir = golden.fir('g48k_k30')
        table = _table(native, ctx, fir)
        xm = rng.uniform(-1, 1, (5, 90002, 1)).astype(np.float32)
        want = c_oracle.convolve(np.ascontiguousarray(np.repeat(xm, 2, axis=2)), *O.fir_to_taps(fir), threads=4)
        for merged in ('1', '0'):
            monkeypatch.setenv('VND_WIN_EXACT_MERGED', merged)
            text = table.describe(5, 90002, 1, d.MODE_EXACT)
            assert text.startswith('conv_spec_exact_window') and ('split-by-channel' in text) == (merged == '0'), text
            assert np.array_equal(table.convolve_host(xm, d.MODE_EXACT), want), merged
        monkeypatch.delenv('VND_WIN_EXACT_MERGED')
        table.close()
        # the class path's table, exact: stereo, and a mono input fanned out
        vn = d.VelvetNoise(sample_rate_hz=48000, seed=1)
        cls = vn._device_table()
        taps = O.generate_class_taps(sample_rate_hz=48000, seed=1)
        xs = rng.uniform(-1, 1, (4, 80000, 2)).astype(np.float32)
        want = np.stack([O.class_convolve(s, taps, (0.85, 0.55, 0.35, 0.2), 2) for s in xs])
        want_m = np.stack([O.class_convolve(np.ascontiguousarray(np.repeat(s[:, :1], 2, axis=1)), taps, (0.85, 0.55, 0.35, 0.2), 2) for s in xs])
        for split in ('1', '0'):
            monkeypatch.setenv('VND_WIN_SPLIT_CLASS', split)
            text = cls.describe(4, 80000, 2, d.MODE_EXACT)
            assert text.startswith('conv_spec_exact_window') and (('frames_per_lane=64 ' in text) == (split == '1') or split == '1'), text
            assert np.array_equal(cls.convolve_host(xs, d.MODE_EXACT), want), split
            assert np.array_equal(cls.convolve_host(np.ascontiguousarray(xs[:, :, :1]), d.MODE_EXACT), want_m), split
    finally:
        ctx.set_variant(-1)


@pytest.mark.parametrize('pool,n', [(7, 3 * 8192 + 50), (13, 8192 * 2), (5, 70000), (3, 200002)])
def test_balanced_cut_equals_the_spans(env, golden, monkeypatch, pool, n):
    """Round 6: pools whose spans would fill the one round of workgroups unevenly are cut into equal contiguous ranges of the pool's
    tiles (vnd_win_kernel.inc: bal_total) - a workgroup's range may end inside a stream and go on in the next one.  Forced here
    (VND_WIN_BALANCE=2) on small ragged pools - ranges of a tile or two, ranges over three streams, tails inside a tile: exact mode
    bit for bit the oracle on every stream, fast mode the very bits of the uniform spans (an output depends on table and position
    alone), function- and class-path tables, a mono input."""
    d, native, ctx = env
    ctx.set_variant(FORCE)
    rng = np.random.default_rng(pool * 1000 + n % 997)
    try:
        fir = golden.fir('g48k_k30')
        table = _table(native, ctx, fir)
        x = rng.uniform(-1, 1, (pool, n, 2)).astype(np.float32)
        want = c_oracle.convolve(x, *O.fir_to_taps(fir), threads=4)
        outs = {}
        for bal in ('2', '0'):
            monkeypatch.setenv('VND_WIN_BALANCE', bal)
            text = table.describe(pool, n, 2, d.MODE_FAST)
            assert ('balanced ranges' in text) == (bal == '2'), text
            outs[bal] = table.convolve_host(x, d.MODE_FAST)
            assert np.array_equal(table.convolve_host(x, d.MODE_EXACT), want), bal
        assert np.array_equal(outs['2'], outs['0']) and _err(outs['2'], want) <= TOL_PEAK
        monkeypatch.setenv('VND_WIN_BALANCE', '2')
        xm = np.ascontiguousarray(x[:, :, :1])
        want_m = c_oracle.convolve(np.ascontiguousarray(np.repeat(xm, 2, axis=2)), *O.fir_to_taps(fir), threads=4)
        assert 'balanced ranges' in table.describe(pool, n, 1, d.MODE_EXACT)
        assert np.array_equal(table.convolve_host(xm, d.MODE_EXACT), want_m)
        table.close()
        cls = d.VelvetNoise(sample_rate_hz=48000, seed=1)._device_table()
        taps = O.generate_class_taps(sample_rate_hz=48000, seed=1)
        want_c = np.stack([O.class_convolve(s, taps, (0.85, 0.55, 0.35, 0.2), 2) for s in x])
        assert 'balanced ranges' in cls.describe(pool, n, 2, d.MODE_EXACT)
        assert np.array_equal(cls.convolve_host(x, d.MODE_EXACT), want_c)
        # ... and under the decorrelate stage (pointwise steps and the RMS block sums in the store phase: a block's index comes from the
        # range's first tile): the generic kernels' bytes
        import torch
        xd = torch.from_numpy(x).cuda()
        ws_bytes = native.decorrelate_workspace_bytes(pool, n, 2)
        outs = {}
        for label, variant in (('balanced', FORCE | (1 << 15)), ('generic', GENERIC)):
            ctx.set_variant(variant)
            yd = torch.empty_like(xd)
            ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
            cls.decorrelate_device(xd.data_ptr(), yd.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, ms_encode=True, width=0.3, normalize=True,
                                   workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs[label] = yd
        assert torch.equal(outs['balanced'], outs['generic'])
    finally:
        ctx.set_variant(-1)
