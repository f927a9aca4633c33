"""GPU tier: the per-table (hipRTC) fast kernel - VND_MODE_FAST's default whenever it applies -
against the oracle, driven through span seams, ring wrap-arounds, stream tails and real sizes.
Bar: <= 1e-6 of the output peak (the north-star tolerance), as for the generic fast kernel."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu

TOL_PEAK = 1e-6
FORCE = 1 << 23            # specialise however little work there is
GENERIC = 1 << 25          # never specialise
EXACT_TOO = 1 << 15        # specialise VND_MODE_EXACT whatever VND_SPEC_EXACT says (it is on by default since the shifted plane copies)


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


def _check(got, want, what, tol=TOL_PEAK):
    peak = float(np.max(np.abs(want))) or 1.0
    err = float(np.max(np.abs(got.astype(np.float64) - want))) / peak
    assert err <= tol, f'{what}: {err:.2e} of peak'


def test_headline_workload_takes_the_specialised_kernel(env, golden):
    d, native, ctx = env
    ctx.set_variant(-1)
    table = _table(native, ctx, golden.fir('g48k_k30'))
    text = table.describe(128, 480000, 2, d.MODE_FAST)
    assert text.startswith('conv_spec'), text            # a silent fallback must not pass for the real thing
    exact = table.describe(128, 480000, 2, d.MODE_EXACT)
    # the window form: 64-frame runs with the waves split over the two channels (function-path tables, exact mode)
    assert exact.startswith('conv_spec_exact_window') and 'frames_per_lane=64 ' in exact and 'waves=split-by-channel' in exact, exact
    ctx.set_variant(1 << 5)                                                  # window form off: the pair-read exact kernel
    exact = table.describe(128, 480000, 2, d.MODE_EXACT)
    assert exact.startswith('conv_spec_exact') and 'tile=1024' in exact, exact       # 256 threads x 2 pairs with the shifted copies
    ctx.set_variant(GENERIC)
    assert table.describe(128, 480000, 2, d.MODE_EXACT).startswith('conv_ordered')
    assert table.describe(128, 480000, 2, d.MODE_FAST).startswith('conv_fast')
    ctx.set_variant(-1)
    assert table.describe(1, 5000, 2, d.MODE_FAST).startswith('conv_fast')       # too little work for persistent workgroups
    table.close()


@pytest.mark.parametrize('gname', ['g48k_k30', 'g44k_k30', 'g48k_k128_l', 'g48k_k128_u', 'g44k_noenv', 'g96k_k64_c8'])
def test_forced_small_signals_and_span_seams(env, golden, gname, monkeypatch):
    """Every length class around the tile (1024 frames) and ring (slots x tile) sizes, spans of 1 to
    7 tiles so that seams, carries and the span-end reduction all run, batches, and both span
    arithmetic extremes.  Checked against the NumPy oracle."""
    d, native, ctx = env
    monkeypatch.setenv('VND_WIN_QUAD', '0')      # the PAIR-READ kernel on the 8-channel table too (6, 10, ... channels take it)
    fir = golden.fir(gname)
    C = fir.shape[1]
    table = _table(native, ctx, fir)
    # 1e-6 of peak: the north star's tolerance, for the 128-tap tables too (2e-6 until round 4)
    tol = TOL_PEAK
    rng = np.random.default_rng(11)
    lengths = [1, 2, 31, 1023, 1024, 1025, 2047, 2048, 4096, 4097, 5000, 9001, 12346]
    for n in lengths:
        for batch in (1, 3):
            if batch > 1 and n % 2:
                continue                                  # odd stereo streams are 8-byte aligned: generic kernel (below)
            x = rng.uniform(-1, 1, (batch, n, C)).astype(np.float32)
            want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
            for min_span, rounds in ((1, 7), (2, 1), (7, 3)):
                ctx.set_variant(FORCE | span_bits(min_span, rounds))
                assert table.describe(batch, n, C, d.MODE_FAST).startswith('conv_spec')
                _check(table.convolve_host(x, d.MODE_FAST), want, f'{gname} n={n} batch={batch} spans=({min_span},{rounds})', tol)
    ctx.set_variant(-1)
    table.close()


def test_other_geometries(env, golden):
    """The tile shapes the geometry choice can fall back to (pairs per lane 1, 2, 8)."""
    d, native, ctx = env
    fir = golden.fir('g48k_k30')
    table = _table(native, ctx, fir)
    x = np.random.default_rng(3).uniform(-1, 1, (2, 30000, 2)).astype(np.float32)
    want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(2)])
    for rr in (1, 2, 8):
        ctx.set_variant(FORCE | span_bits(1, 5) | rr)
        text = table.describe(2, 30000, 2, d.MODE_FAST)
        assert f'pairs_per_lane={rr} ' in text, text
        _check(table.convolve_host(x, d.MODE_FAST), want, f'rr={rr}')
    for dd in (1, 2):                                    # prefetch depth must divide the slot count: 4 slots at 2 pairs per lane
        ctx.set_variant(FORCE | span_bits(1, 5) | (dd << 26) | 2)
        text = table.describe(2, 30000, 2, d.MODE_FAST)
        assert f'prefetch={dd} ' in text and 'ring_slots=4' in text, text
        _check(table.convolve_host(x, d.MODE_FAST), want, f'dd={dd}')
    ctx.set_variant(-1)
    table.close()


def test_out_of_scope_launches_take_the_generic_kernel(env, golden):
    d, native, ctx = env
    ctx.set_variant(FORCE)
    t3 = _table(native, ctx, golden.fir('g48k_c3'))                       # odd channel count
    assert t3.describe(4, 50000, 3, d.MODE_FAST).startswith('conv_fast')
    t2 = _table(native, ctx, golden.fir('g48k_k30'))
    assert t2.describe(3, 50001, 2, d.MODE_FAST).startswith('conv_fast')  # odd streams: 8-byte aligned bases
    assert t2.describe(4, 50000, 1, d.MODE_FAST).startswith('conv_spec')  # a mono input through the stereo table is in scope (VS_BC) ...
    assert t2.describe(4, 50001, 1, d.MODE_FAST).startswith('conv_fast')  # ... unless its streams start off the 8-byte grid
    t8 = _table(native, ctx, golden.fir('g96k_k64_c8'))
    assert t8.describe(4, 50000, 2, d.MODE_FAST).startswith('conv_fast')  # wider fan-outs are not
    t8.close()
    x = np.random.default_rng(5).uniform(-1, 1, (3, 50001, 2)).astype(np.float32)
    want = np.stack([O.convolve_velvet_noise(x[b], golden.fir('g48k_k30')) for b in range(3)])
    _check(t2.convolve_host(x, d.MODE_FAST), want, 'odd streams')
    # a base address that is not 16-byte aligned: the per-table kernel declines, the generic one runs on dword accesses
    import torch
    flat = torch.empty(4 * 60000 * 2 + 8, dtype=torch.float32, device='cuda').uniform_(-1, 1)
    out = torch.empty_like(flat)
    xs, ys = flat[1:1 + 4 * 60000 * 2], out[1:1 + 4 * 60000 * 2]
    assert xs.data_ptr() % 16 == 4
    t2.convolve_device(xs.data_ptr(), ys.data_ptr(), 4, 60000, 2, mode=d.MODE_FAST, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    xh = xs.cpu().numpy().reshape(4, 60000, 2)
    want = np.stack([O.convolve_velvet_noise(xh[b], golden.fir('g48k_k30')) for b in range(4)])
    _check(ys.cpu().numpy().reshape(4, 60000, 2), want, 'misaligned base')
    ctx.set_variant(-1)
    t2.close(); t3.close()


def test_real_size_pool_matches_the_exact_kernel(env, golden):
    """cfg2 streams at full length, enough of them for the automatic choice to specialise (8 spans
    of 59 tiles per stream): every stream against the bit-exact kernel on the device, three of them
    against the C oracle, and batch == loop."""
    import torch
    d, native, ctx = env
    ctx.set_variant(-1)
    fir = golden.fir('g48k_k30')
    table = _table(native, ctx, fir)
    pool, n = 32, 480000
    assert table.describe(pool, n, 2, d.MODE_FAST).startswith('conv_spec')
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y, ye = torch.empty_like(x), torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=d.MODE_FAST, stream=stream)
    table.convolve_device(x.data_ptr(), ye.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, stream=stream)
    torch.cuda.synchronize()
    peak = float(ye.abs().max())
    assert float((y - ye).abs().max()) <= TOL_PEAK * peak
    offs, idx, w = O.fir_to_taps(fir)
    for b in (0, 11, 31):
        want = c_oracle.convolve(x[b].cpu().numpy(), offs, idx, w)
        assert np.array_equal(ye[b].cpu().numpy(), want)
        _check(y[b].cpu().numpy(), want, f'stream {b}')
    # the same streams one launch each (forced: a single stream is too little work otherwise)
    ctx.set_variant(FORCE)
    y1 = torch.empty((1, n, 2), dtype=torch.float32, device='cuda')
    for b in (5, 31):
        table.convolve_device(x[b].data_ptr(), y1.data_ptr(), 1, n, 2, mode=d.MODE_FAST, stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(y1[0], y[b])                  # same arithmetic whatever the span layout: bit-identical
    ctx.set_variant(-1)
    table.close()


def test_headline_span_layout_past_four_gib_of_offsets(env, golden):
    """The bench's headline launch shape - one span of 59 tiles per stream, several units per workgroup - on a pool whose arrays pass
    4 GiB (1536 cfg2 signals: 5.9 GB each way): stream offsets beyond 32 bits, descriptor bases recomputed per unit.  Every stream fast
    against exact on the device; the first, a middle and the LAST stream (the highest addresses) of both against the C oracle - exact bit
    for bit."""
    import torch
    d, native, ctx = env
    ctx.set_variant(-1)
    fir = golden.fir('g48k_k30')
    table = _table(native, ctx, fir)
    pool, n = 1536, 480000                                 # (a multiple of the 512 resident workgroups: one span per stream, as the bench's 2048)
    assert pool * n * 2 * 4 > 1 << 32
    for mode in (d.MODE_FAST, d.MODE_EXACT):
        text = table.describe(pool, n, 2, mode)
        assert text.startswith('conv_spec') and '_window' in text and '1 spans x 59 tiles per stream' in text, text
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y, ye = torch.empty_like(x), torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=d.MODE_FAST, stream=stream)
    table.convolve_device(x.data_ptr(), ye.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, stream=stream)
    torch.cuda.synchronize()
    per_stream = torch.cat([(y[b0:b0 + 96] - ye[b0:b0 + 96]).abs().amax(dim=(1, 2)) for b0 in range(0, pool, 96)])
    peak = float(ye[::97].abs().max())
    assert float(per_stream.max()) <= TOL_PEAK * peak, f'stream {int(per_stream.argmax())}: {float(per_stream.max()) / peak:.2e} of peak'
    offs, idx, w = O.fir_to_taps(fir)
    for b in (0, 767, pool - 1):
        want = c_oracle.convolve(x[b].cpu().numpy(), offs, idx, w, threads=8)
        assert np.array_equal(ye[b].cpu().numpy(), want), f'exact kernel, stream {b}'
        _check(y[b].cpu().numpy(), want, f'stream {b}')
    del x, y, ye
    torch.cuda.empty_cache()
    table.close()


def test_linearity_and_shift_at_full_size(env, golden):
    """Size-independent properties on the specialised path: exact dyadic scaling and shift invariance."""
    import torch
    d, native, ctx = env
    ctx.set_variant(-1)
    table = _table(native, ctx, golden.fir('g48k_k30'))
    pool, n = 64, 240000
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    assert table.describe(pool, n, 2, d.MODE_FAST).startswith('conv_spec')
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=d.MODE_FAST, stream=s)
    x4 = (x * 4.0).contiguous()
    table.convolve_device(x4.data_ptr(), y2.data_ptr(), pool, n, 2, mode=d.MODE_FAST, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(y2, y * 4.0)
    # shifting the input by one tile shifts the output by one tile (interior frames)
    shift = 2048
    xs = torch.zeros_like(x)
    xs[:, shift:] = x[:, :-shift]
    table.convolve_device(xs.data_ptr(), y2.data_ptr(), pool, n, 2, mode=d.MODE_FAST, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(y2[:, shift:-2000], y[:, :-shift - 2000])
    table.close()


# ---- VND_MODE_EXACT through the specialised kernel: bit-identical, like the generic ordered kernel ----
@pytest.mark.parametrize('gname', ['g48k_k30', 'g44k_k30', 'g48k_k128_l', 'g44k_noenv', 'g96k_k64_c8'])
def test_exact_mode_specialised_is_bit_identical(env, golden, gname, monkeypatch):
    """Table order, separately rounded products and sums, odd offsets as two dword reads: the per-table
    kernel in exact mode must equal the oracle bit for bit - through span seams, ring wrap-arounds,
    stream tails and batches, as the fast mode's test above."""
    d, native, ctx = env
    monkeypatch.setenv('VND_WIN_QUAD', '0')      # the pair-read kernel on the 8-channel table too
    fir = golden.fir(gname)
    C = fir.shape[1]
    table = _table(native, ctx, fir)
    rng = np.random.default_rng(21)
    for n in [1, 2, 31, 1023, 1024, 1025, 1536, 3071, 4097, 9001, 12346]:
        for batch in (1, 3):
            if batch > 1 and n % 2:
                continue
            x = rng.uniform(-1, 1, (batch, n, C)).astype(np.float32)
            want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(batch)])
            for min_span, rounds in ((1, 7), (3, 2)):
                ctx.set_variant(FORCE | EXACT_TOO | span_bits(min_span, rounds))
                assert table.describe(batch, n, C, d.MODE_EXACT).startswith('conv_spec_exact')
                got = table.convolve_host(x, d.MODE_EXACT)
                assert np.array_equal(got, want), f'{gname} n={n} batch={batch} spans=({min_span},{rounds})'
    ctx.set_variant(-1)
    table.close()


@pytest.mark.parametrize('name', sorted(__import__('json').loads((__import__('pathlib').Path(__file__).parent / 'golden' / 'manifest.json').read_text())['cls_convolve']))
def test_exact_mode_specialised_class_path(env, golden, name):
    """VelvetNoise.convolve's association (segments of -/+ unit taps, one multiply per segment, segments
    summed; decorrelation.py:402-414) through the specialised exact kernel: the reference's sha256."""
    from conftest import make_input
    d, native, ctx = env
    meta = golden.manifest['cls_convolve'][name]
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps'][meta['class']]['kwargs'].items()}
    vn = d.VelvetNoise(**kw)
    x = make_input(meta['input'])
    ctx.set_variant(FORCE | EXACT_TOO | span_bits(1, 3))
    try:
        y = vn.convolve(x)
        # a table whose channels are all filtered really takes the per-table kernel (one with a copied-through
        # channel is outside its scope and comes out of the generic kernel)
        every = len(kw.get('filtered_channels', (0, 1))) == vn.num_outs and vn.num_outs % 2 == 0 and \
            vn._device_table().max_index < 3000                      # (a 0.5 s filter's history does not fit the LDS ring)
        launch = vn._device_table().describe(1, len(x), vn.num_outs, d.MODE_EXACT)
        assert launch.startswith('conv_spec_exact') == every, (name, launch)
    finally:
        ctx.set_variant(-1)
    golden.expect(name, y, exact=x.dtype == np.float32, rtol_peak=TOL_PEAK)


def test_exact_mode_real_size_equals_the_generic_kernel(env, golden):
    import torch
    d, native, ctx = env
    fir = golden.fir('g48k_k30')
    table = _table(native, ctx, fir)
    pool, n = 32, 480000
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y, yg = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    ctx.set_variant(EXACT_TOO)
    assert table.describe(pool, n, 2, d.MODE_EXACT).startswith('conv_spec_exact')
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, stream=s)
    ctx.set_variant(GENERIC)
    assert table.describe(pool, n, 2, d.MODE_EXACT).startswith('conv_ordered')
    table.convolve_device(x.data_ptr(), yg.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, stream=s)
    torch.cuda.synchronize()
    ctx.set_variant(-1)
    assert torch.equal(y, yg)
    offs, idx, w = O.fir_to_taps(fir)
    assert np.array_equal(y[7].cpu().numpy(), c_oracle.convolve(x[7].cpu().numpy(), offs, idx, w))
    table.close()


def test_class_path_exact_takes_the_specialised_kernel_by_default(env, golden):
    """VelvetNoise.convolve's table (every weight +-1: one packed add per tap) gains 29 % from the per-table kernel in
    VND_MODE_EXACT, so a large enough batch takes it WITHOUT the opt-in; bit-identical to the generic ordered kernel
    and, on one stream, to the oracle's class-path association."""
    import torch
    d, native, ctx = env
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps']['v48k']['kwargs'].items()}
    vn = d.VelvetNoise(**kw)
    table = vn._device_table()
    pool, n = 32, 480000
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y, yg = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    ctx.set_variant(-1)
    assert table.describe(pool, n, 2, d.MODE_EXACT).startswith('conv_spec_exact')
    table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, stream=s)
    ctx.set_variant(GENERIC)
    assert table.describe(pool, n, 2, d.MODE_EXACT).startswith('conv_ordered')
    table.convolve_device(x.data_ptr(), yg.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, stream=s)
    torch.cuda.synchronize()
    ctx.set_variant(-1)
    assert torch.equal(y, yg)
    taps = golden.class_taps('v48k', 2)
    env_gains = tuple(golden.manifest['class_taps']['v48k']['envelope'])
    assert np.array_equal(y[5].cpu().numpy(), O.class_convolve(x[5].cpu().numpy(), taps, env_gains, 2))
    # so does a function-path table (arbitrary weights: multiply + add per tap) since the shifted plane copies
    fn = _table(native, ctx, golden.fir('g48k_k30'))
    assert fn.describe(pool, n, 2, d.MODE_EXACT).startswith('conv_spec_exact')
    fn.close()


@pytest.mark.parametrize('ms_encode,width', [(True, None), (True, 0.3), (False, 0.75)])
def test_decorrelate_stage_through_the_specialised_kernel(env, golden, ms_encode, width, tmp_path, monkeypatch):
    """The whole exact stage on a batch: the side-channel encode and stereo width ride in the per-table exact
    kernel's store phase (VS_EPI) exactly as in the generic ordered kernel's - same bytes out, and the
    reference's own stage (oracle) on one stream."""
    import torch
    d, native, ctx = env
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps']['v48k']['kwargs'].items()}
    kw['seed'] = 1000 + int(ms_encode) * 2 + int(width is not None) + int(100 * (width or 0))      # a table no other test has built
    vn = d.VelvetNoise(**kw)
    table = vn._device_table()
    # proof that the per-table kernel with the fused steps is what runs: a fresh kernel cache receives exactly one
    # code object, and the generated source says VS_EPI 1 / VS_EXACT 1
    monkeypatch.setenv('VND_SPEC_CACHE_DIR', str(tmp_path / 'cache'))
    monkeypatch.setenv('VND_SPEC_DUMP', str(tmp_path / 'kernel.hip'))
    pool, n = 24, 300002                    # (an odd frame count puts every second stream off the 16-byte grid: generic kernel)
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    s = torch.cuda.current_stream().cuda_stream
    ws_bytes = native.decorrelate_workspace_bytes(pool, n, 2)
    outs = {}
    for label, variant in (('spec', FORCE | span_bits(1, 2)), ('generic', GENERIC)):
        ctx.set_variant(variant)
        y = torch.empty_like(x)
        ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
        table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=d.MODE_EXACT, ms_encode=ms_encode, width=width,
                                 normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=s)
        torch.cuda.synchronize()
        outs[label] = y
    ctx.set_variant(-1)
    assert len(list((tmp_path / 'cache').glob('*.co'))) == 1
    source = (tmp_path / 'kernel.hip').read_text()
    # (a stereo table's exact mode takes the WINDOW form of the per-table kernel too: the same switches, prefixed VW_)
    assert '#define VW_EPI 1' in source and '#define VW_EXACT 1' in source
    assert torch.equal(outs['spec'], outs['generic'])
    want = O.decorrelate(x[3].cpu().numpy().copy(), sample_rate_hz=48000, seed=kw['seed'], width=width, mode='MS' if ms_encode else 'LR')
    assert np.array_equal(outs['spec'][3].cpu().numpy(), want)


@pytest.mark.parametrize('which', ['function_path', 'class_path'])
def test_mono_fan_out_through_the_specialised_kernels(env, golden, which):
    """A mono input through a stereo table (mono -> stereo decorrelation, decorrelation.py:431-432): the per-table
    kernels stage ONE plane that both channels' taps read (VS_BC).  Fast mode within tolerance, exact mode bit for bit,
    against the oracle on the replicated input; span seams and a ragged last tile included."""
    d, native, ctx = env
    pool, n = 20, 100002
    x = np.random.default_rng(31).uniform(-1, 1, (pool, n, 1)).astype(np.float32)
    stereo = np.ascontiguousarray(np.repeat(x, 2, axis=2))
    if which == 'function_path':
        fir = golden.fir('g48k_k30')
        table = _table(native, ctx, fir)
        offs, idx, w = O.fir_to_taps(fir)
        want = c_oracle.convolve(stereo, offs, idx, w, threads=4)
    else:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps']['v48k']['kwargs'].items()}
        table = d.VelvetNoise(**kw)._device_table()
        taps = golden.class_taps('v48k', 2)
        env_gains = tuple(golden.manifest['class_taps']['v48k']['envelope'])
        want = np.stack([O.class_convolve(s, taps, env_gains, 2) for s in stereo])
    try:
        for spans in ((1, 2), (1, 5)):
            ctx.set_variant(FORCE | EXACT_TOO | span_bits(*spans))
            for mode in (d.MODE_FAST, d.MODE_EXACT):
                launch = table.describe(pool, n, 1, mode)
                assert launch.startswith('conv_spec'), launch
                if which == 'function_path':
                    # the window forms - both modes the plain 32-frame form with ONE read stream for both output channels (their taps lie
                    # almost alike: win_taps_function_merged; exact since round 6: win_taps_function_exact_merged, the products shared
                    # too), 16-byte mono loads
                    assert '_window' in launch and 'frames_per_lane=32' in launch and 'split-by-channel' not in launch, launch
                else:
                    # class-path tables: fast as above; exact (a pass per segment and sign list) in the split form - 64-frame runs, a
                    # wave per OUTPUT channel, the input staged into both plane sets (vw_span_s, VW_BC)
                    assert '_window' in launch and ('frames_per_lane=32' in launch if mode == d.MODE_FAST else 'waves=split-by-channel' in launch), launch
                y = table.convolve_host(x, mode)
                if mode == d.MODE_EXACT:
                    assert np.array_equal(y, want), (which, spans)
                else:
                    assert np.max(np.abs(y - want)) <= TOL_PEAK * np.max(np.abs(want)), (which, spans)
    finally:
        ctx.set_variant(-1)
    if which == 'function_path':
        table.close()


def test_mono_decorrelate_stage_through_the_specialised_kernel(env, golden, tmp_path, monkeypatch):
    """VelvetNoise.decorrelate of MONO signals in a batch: fan-out, side-channel encode and width in the per-table exact
    kernel's store phase - the same bytes as the generic kernels, and the oracle's stage on one signal."""
    import torch
    d, native, ctx = env
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps']['v48k']['kwargs'].items()}
    kw['seed'] = 2024
    table = d.VelvetNoise(**kw)._device_table()
    monkeypatch.setenv('VND_SPEC_CACHE_DIR', str(tmp_path / 'cache'))
    monkeypatch.setenv('VND_SPEC_DUMP', str(tmp_path / 'kernel.hip'))
    pool, n = 24, 200002
    x = torch.empty((pool, n, 1), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    s = torch.cuda.current_stream().cuda_stream
    ws_bytes = native.decorrelate_workspace_bytes(pool, n, 2)
    outs, sources = {}, {}
    # window: the plain 32-frame window form (VW_BC + VW_EPI: both channels' passes read the one plane set, the store phase encodes with the
    # mono frame and leaves the block sums of the exact RMS - the default since round 6); pair_read: the form it replaced (VND_WIN_FANOUT_EPI=0)
    for label, variant, fan in (('window', FORCE | EXACT_TOO | span_bits(1, 2), '1'), ('pair_read', FORCE | EXACT_TOO | span_bits(1, 2), '0'), ('generic', GENERIC, '1')):
        ctx.set_variant(variant)
        monkeypatch.setenv('VND_WIN_FANOUT_EPI', fan)
        y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
        ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
        table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, 1, mode=d.MODE_EXACT, ms_encode=True, width=0.4,
                                 normalize=True, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=s)
        torch.cuda.synchronize()
        outs[label] = y
        if label != 'generic':
            sources[label] = (tmp_path / 'kernel.hip').read_text()
    ctx.set_variant(-1)
    assert '#define VW_EPI 1' in sources['window'] and '#define VW_BC 1' in sources['window'] and '#define VW_EXACT 1' in sources['window']
    assert '#define VS_EPI 1' in sources['pair_read'] and '#define VS_BC 1' in sources['pair_read']
    assert len(list((tmp_path / 'cache').glob('*.co'))) == 2
    assert torch.equal(outs['window'], outs['generic']) and torch.equal(outs['pair_read'], outs['generic'])
    outs['spec'] = outs['window']
    want = O.decorrelate(x[5, :, 0].cpu().numpy().copy(), sample_rate_hz=48000, seed=2024, width=0.4, mode='MS')
    assert np.array_equal(outs['spec'][5].cpu().numpy(), want)


@pytest.mark.parametrize('mono', [False, True])
def test_fast_decorrelate_stage_through_the_specialised_kernel(env, golden, tmp_path, monkeypatch, mono):
    """The throughput mode's stage (reference-order RMS sums, the default normaliser flag): side-channel encode and width
    in the per-table FAST kernel's store phase - the tile's last frame is completed one tile later, so its input frame
    is carried along with its even part.  Within the fast mode's tolerance of the oracle's stage; span seams included."""
    import torch
    d, native, ctx = env
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps']['v48k']['kwargs'].items()}
    kw['seed'] = 3000 + int(mono)
    table = d.VelvetNoise(**kw)._device_table()
    monkeypatch.setenv('VND_SPEC_CACHE_DIR', str(tmp_path / 'cache'))
    monkeypatch.setenv('VND_SPEC_DUMP', str(tmp_path / 'kernel.hip'))
    pool, n, cx = 24, 200002, (1 if mono else 2)
    x = torch.empty((pool, n, cx), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    s = torch.cuda.current_stream().cuda_stream
    ws_bytes = native.decorrelate_workspace_bytes(pool, n, 2)
    ctx.set_variant(FORCE | span_bits(1, 3))
    y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
    ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
    table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, cx, mode=d.MODE_FAST, ms_encode=True, width=0.6,
                             normalize=2, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=s)
    torch.cuda.synchronize()
    ctx.set_variant(-1)
    source = (tmp_path / 'kernel.hip').read_text()
    # (the fast mode takes the WINDOW form of the per-table kernel - the same switches, prefixed VW_ - for a stereo input and,
    #  since round 4, for a mono input fanned out: the plain form, one read stream for both output channels)
    px = 'VW'
    assert f'#define {px}_EPI 1' in source and f'#define {px}_EXACT 0' in source and f'#define {px}_BC {int(mono)}' in source
    for b in (0, 11, 23):
        sig = x[b].cpu().numpy()
        want = O.decorrelate((sig[:, 0] if mono else sig).copy(), sample_rate_hz=48000, seed=kw['seed'], width=0.6, mode='MS')
        got = y[b].cpu().numpy()
        assert np.max(np.abs(got - want)) <= 3e-6 * np.max(np.abs(want)), (b, mono)


def test_small_launches_never_trigger_a_build(env, golden, tmp_path, monkeypatch):
    """A launch too small to be worth a hipRTC build (1.5-5 s) takes the per-table kernel only when its code object already
    exists - built by a larger launch of the same geometry, or found in the disk cache - and the generic kernel otherwise."""
    import torch
    d, native, ctx = env
    monkeypatch.setenv('VND_SPEC_CACHE_DIR', str(tmp_path / 'cache'))
    fir = d.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=77)
    table = _table(native, ctx, fir)
    pool, n = 8, 480000                                                   # 3.8 M frames: in scope, but below the build threshold
    x = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    ctx.set_variant(-1)
    for mode, generic_name in ((d.MODE_EXACT, 'conv_ordered'), (d.MODE_FAST, 'conv_fast')):
        assert table.describe(pool, n, 2, mode).startswith(generic_name)
        table.convolve_device(x.data_ptr(), y.data_ptr(), pool, n, 2, mode=mode, stream=s)
        torch.cuda.synchronize()
        assert not list((tmp_path / 'cache').glob('*.co')) or mode == d.MODE_FAST          # nothing was built for it
        ctx.set_variant(FORCE)                                            # what a larger launch would have done: build it
        table.convolve_device(x.data_ptr(), y2.data_ptr(), pool, n, 2, mode=mode, stream=s)
        torch.cuda.synchronize()
        ctx.set_variant(-1)
        assert table.describe(pool, n, 2, mode).startswith('conv_spec')  # now the small launch takes it
        table.convolve_device(x.data_ptr(), y2.data_ptr(), pool, n, 2, mode=mode, stream=s)
        torch.cuda.synchronize()
        if mode == d.MODE_EXACT:
            assert torch.equal(y, y2)
        else:
            assert float((y - y2).abs().max()) <= TOL_PEAK * float(y.abs().max())
    # a second table object with the same taps finds the code objects in the disk cache
    again = _table(native, ctx, fir)
    assert again.describe(pool, n, 2, d.MODE_EXACT).startswith('conv_spec_exact')
    table.close(); again.close()


def test_failed_runtime_build_falls_back_to_the_generic_kernel(env, golden, monkeypatch):
    """If hipRTC cannot build the per-table kernel (here: an injected #error), the launch silently takes the
    generic HIP kernel - never a CPU path - the result is still right, and vnd_describe_launch says so."""
    d, native, ctx = env
    fir = golden.fir('g44k_55ms')                        # a table no other test has built a module for
    monkeypatch.setenv('VND_SPEC_BREAK', '1')
    table = _table(native, ctx, fir)
    ctx.set_variant(FORCE)
    try:
        x = np.random.default_rng(8).uniform(-1, 1, (2, 40000, 2)).astype(np.float32)
        want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(2)])
        got = table.convolve_host(x, d.MODE_FAST)
        _check(got, want, 'fallback')
        assert table.describe(2, 40000, 2, d.MODE_FAST).startswith('conv_fast')
    finally:
        ctx.set_variant(-1)
        table.close()
    monkeypatch.delenv('VND_SPEC_BREAK')
    table = _table(native, ctx, fir)                     # a new table builds its module afresh
    ctx.set_variant(FORCE)
    try:
        assert table.describe(2, 40000, 2, d.MODE_FAST).startswith('conv_spec')
    finally:
        ctx.set_variant(-1)
        table.close()


@pytest.mark.parametrize('name', sorted(__import__('json').loads((__import__('pathlib').Path(__file__).parent / 'golden' / 'manifest.json').read_text())['cls_convolve']))
def test_fast_mode_specialised_class_path(env, golden, name):
    """Class-path tables (segment gains folded into the weights, duplicate indices summed) through the
    per-table fast kernel: the reference's VelvetNoise.convolve within the fast mode's tolerance; tables
    with a pass-through channel are outside its scope and must still come out right (generic kernel)."""
    from conftest import make_input
    d, native, ctx = env
    meta = golden.manifest['cls_convolve'][name]
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps'][meta['class']]['kwargs'].items()}
    vn = d.VelvetNoise(**kw)
    x = make_input(meta['input'])
    ctx.set_variant(FORCE | span_bits(1, 3))
    d.set_default_mode(d.MODE_FAST)
    try:
        y = vn.convolve(x)
        every = len(kw.get('filtered_channels', (0, 1))) == vn.num_outs and vn.num_outs % 2 == 0 and \
            vn._device_table().max_index < 3000
        launch = vn._device_table().describe(1, len(x), vn.num_outs, d.MODE_FAST)
        assert launch.startswith('conv_spec') == every, (name, launch)
    finally:
        d.set_default_mode(d.MODE_EXACT)
        ctx.set_variant(-1)
    # 1. The north star's bar, against the FUNCTION-path oracle of the same effective table (every tap with its segment gain, ascending
    #    index, duplicates kept as separate taps; `acc += x * w` in float32): <= 1e-6 of peak, 128-tap tables included.
    a = vn._tap_arrays()
    offs, idx, w = [0], [], []
    seg_of_tap = np.searchsorted(a.seg_end, np.arange(len(a.tap_index)), side='right')
    for c in range(vn.num_outs):
        lo, hi = int(a.tap_offsets[c]), int(a.tap_offsets[c + 1])
        order = lo + np.argsort(a.tap_index[lo:hi], kind='stable')
        idx.append(a.tap_index[order])
        w.append((a.tap_weight[order] * a.seg_gain[seg_of_tap[order]]).astype(np.float32))
        offs.append(offs[-1] + hi - lo)
    x32 = np.ascontiguousarray(x if x.ndim == 2 else x[:, None], dtype=np.float32)
    filtered = [c for c in range(vn.num_outs) if a.tap_offsets[c + 1] > a.tap_offsets[c]]
    if x.dtype == np.float32 and x32.shape[1] >= vn.num_outs and filtered:
        want = c_oracle.convolve(np.ascontiguousarray(x32[:, :vn.num_outs]), np.asarray(offs, np.int32), np.concatenate(idx).astype(np.int32),
                                 np.concatenate(w), threads=8)
        peak = float(np.max(np.abs(want[:, filtered]))) or 1.0        # (a one-frame signal: every tap reads past its end)
        err = float(np.max(np.abs(y[:, filtered].astype(np.float64) - want[:, filtered]))) / peak
        assert err <= TOL_PEAK, f'{name}: {err:.3e} of peak from the function-path oracle of the same table'
    # 2. Against the reference's own VelvetNoise.convolve output (the golden fixture): its association - (sum of -x, then +x) * gain per
    #    segment - differs from the function path's by up to 1.2e-6 of peak on the 128-tap tables by itself (the reference's two
    #    paths agree to ~1e-6 only), so this bar is the sum of the two: 2e-6 there, 1e-6 on the 30-tap tables.
    golden.expect(name, y, exact=False, rtol_peak=2e-6 if 'k128' in name else TOL_PEAK)


def test_lopsided_tables(env):
    """A channel without taps, a channel with one tap, taps at index 0 and at the very end of a long filter."""
    d, native, ctx = env
    rng = np.random.default_rng(31)
    x = rng.uniform(-1, 1, (2, 20000, 2)).astype(np.float32)
    firs = []
    f = np.zeros((1500, 2), np.float32); f[[0, 3, 700, 1499], 0] = [0.5, -0.25, 1.0, 0.125]; firs.append(f)          # channel 1 silent
    f = np.zeros((1500, 2), np.float32); f[0, 0] = 1.0; f[1, 1] = -1.0; firs.append(f)                              # identity and a one-frame advance
    f = np.zeros((6000, 2), np.float32); f[[1, 5999], 0] = [1.0, 0.5]; f[[2, 5998], 1] = [0.25, -0.5]; firs.append(f)  # a 5999-frame halo
    for k, fir in enumerate(firs):
        table = _table(native, ctx, fir)
        want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(2)])
        ctx.set_variant(FORCE | span_bits(1, 4))
        # (the 5999-frame halo does not fit the pair-read kernel's ring of under 64 KiB - its exact mode takes the generic
        #  kernel - but the window form's ring, tile + halo in up to 160 KiB of LDS, holds it)
        assert table.describe(2, x.shape[1], 2, d.MODE_FAST).startswith('conv_spec_window'), k
        _check(table.convolve_host(x, d.MODE_FAST), want, f'lopsided table {k}')
        ctx.set_variant(FORCE | EXACT_TOO | span_bits(1, 4))
        assert np.array_equal(table.convolve_host(x, d.MODE_EXACT), want), k
        ctx.set_variant(-1)
        table.close()


def test_compiled_kernels_are_cached_on_disk(env, golden, tmp_path, monkeypatch):
    """The code object of a table's kernel is written under VND_SPEC_CACHE_DIR and loaded from there by the
    next table object with the same content (the second build must not need hipRTC: VND_SPEC_BREAK would fail it)."""
    d, native, ctx = env
    monkeypatch.setenv('VND_SPEC_CACHE_DIR', str(tmp_path / 'cache'))
    fir = golden.fir('g44k_20ms')
    x = np.random.default_rng(2).uniform(-1, 1, (2, 30000, 2)).astype(np.float32)
    want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(2)])
    ctx.set_variant(FORCE)
    try:
        t1 = _table(native, ctx, fir)
        _check(t1.convolve_host(x, d.MODE_FAST), want, 'first build')
        files = list((tmp_path / 'cache').glob('*.co'))
        assert len(files) == 1 and files[0].stat().st_size > 1000, (files, t1.describe(2, 30000, 2, d.MODE_FAST))
        t1.close()
        t2 = _table(native, ctx, fir)
        assert t2.describe(2, 30000, 2, d.MODE_FAST).startswith('conv_spec')
        _check(t2.convolve_host(x, d.MODE_FAST), want, 'from the cache')
        assert len(list((tmp_path / 'cache').glob('*.co'))) == 1
        t2.close()
    finally:
        ctx.set_variant(-1)


@pytest.mark.parametrize('kind', ['uniform', 'int16', 'sparse'])
@pytest.mark.parametrize('cx', [1, 2])
@pytest.mark.parametrize('ms_encode,width,normalize', [(True, None, True), (False, 0.3, True), (True, 0.7, False), (False, None, True)])
def test_exact_stage_in_the_window_form_equals_the_generic_kernels(env, golden, tmp_path, monkeypatch, kind, cx, ms_encode, width, normalize):
    """The exact `decorrelate` stage in the plain window form with VW_EPI - the store phase applies the pointwise steps and leaves the
    block sums of the NumPy-order RMS - for stereo input and (round 6) for MONO input (VW_BC: the mono frame feeds the side-channel
    encode) and for the normaliser alone (LR mode: no pointwise step, the launch is taken for its block sums).  Every combination of
    the stage's steps, ragged lengths across tile and block seams, signals with ties and zero runs in the sums: the same bytes as the
    generic kernels' stage, and the oracle's on one stream."""
    import torch
    d, native, ctx = env
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in golden.manifest['class_taps']['v48k']['kwargs'].items()}
    kw['seed'] = 77
    table = d.VelvetNoise(**kw)._device_table()
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((kind, cx, ms_encode, width, normalize)).encode()))
    monkeypatch.setenv('VND_SPEC_CACHE_DIR', str(tmp_path / 'cache'))
    monkeypatch.setenv('VND_SPEC_DUMP', str(tmp_path / 'kernel.hip'))
    s = torch.cuda.current_stream().cuda_stream
    for pool, n in ((6, 3 * 8192 + 2048 + 78), (3, 2 * 8192)):          # (even lengths: odd mono streams are not 8-byte aligned - generic kernels)
        if kind == 'uniform':
            xh = rng.uniform(-1, 1, (pool, n, cx))
        elif kind == 'int16':
            xh = rng.integers(-32768, 32767, (pool, n, cx)) / 32768.0
        else:
            xh = rng.integers(-3, 4, (pool, n, cx)) * (rng.random((pool, n, cx)) < 0.2)
        x = torch.from_numpy(np.ascontiguousarray(xh, np.float32)).cuda()
        ws_bytes = native.decorrelate_workspace_bytes(pool, n, 2)
        outs = {}
        for label, variant in (('window', FORCE | EXACT_TOO | span_bits(1, 2)), ('generic', GENERIC)):
            ctx.set_variant(variant)
            y = torch.empty((pool, n, 2), dtype=torch.float32, device='cuda')
            ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device='cuda')
            table.decorrelate_device(x.data_ptr(), y.data_ptr(), pool, n, cx, mode=d.MODE_EXACT, ms_encode=ms_encode, width=width,
                                     normalize=normalize, workspace_ptr=ws.data_ptr(), workspace_bytes=ws_bytes, stream=s)
            torch.cuda.synchronize()
            outs[label] = y
        ctx.set_variant(-1)
        source = (tmp_path / 'kernel.hip').read_text()
        if normalize:          # (the block sums bring a mono launch to the window form - with or without pointwise steps; without a normaliser the pair-read form keeps it)
            assert '#define VW_EPI 1' in source and f'#define VW_BC {2 - cx}' in source and '#define VW_EXACT 1' in source
        assert torch.equal(outs['window'], outs['generic']), (pool, n)
        if ms_encode and normalize:
            xs = x[pool - 1].cpu().numpy().copy()
            want = O.decorrelate(xs[:, 0] if cx == 1 else xs, sample_rate_hz=48000, seed=77, width=width, mode='MS')
            assert np.array_equal(outs['window'][pool - 1].cpu().numpy(), want, equal_nan=True)
