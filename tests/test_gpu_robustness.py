"""GPU tier: re-entrancy and multi-context behaviour of the C ABI (VERDICT r1 robustness items)."""
import threading

import numpy as np
import pytest

from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vnd():
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    _native.default_context()
    return d


def test_two_threads_share_the_default_context(vnd, golden):
    """convolve_velvet_noise is re-entrant upstream (decorrelation.py:630-660).  Two Python threads
    (ctypes drops the GIL) push differently sized signals through the ONE default context - whose
    staging buffers get reallocated as the sizes change - and every result must be the oracle's."""
    firs = [golden.fir('g48k_k30'), golden.fir('g44k_k30')]
    rng = np.random.default_rng(5)
    sizes = [1000, 48000, 777, 200000, 31, 96000, 4097, 150000]
    work = [(rng.uniform(-1, 1, (n, 2)).astype(np.float32), firs[i % 2]) for i, n in enumerate(sizes)]
    want = [O.convolve_velvet_noise(x, f) for x, f in work]
    vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=3)
    want_dec = [O.decorrelate(x, sample_rate_hz=48000, seed=3) for x, _ in work]
    failures = []

    def run(order):
        try:
            for _ in range(6):
                for i in order:
                    x, f = work[i]
                    if not np.array_equal(vnd.convolve_velvet_noise(x, f, mode=vnd.MODE_EXACT), want[i]):
                        failures.append(('convolve', i))
                    if not np.array_equal(vn.decorrelate(x), want_dec[i]):
                        failures.append(('decorrelate', i))
        except Exception as e:      # noqa: BLE001
            failures.append(repr(e))

    threads = [threading.Thread(target=run, args=(list(range(8)),)),
               threading.Thread(target=run, args=(list(range(7, -1, -1)),)),
               threading.Thread(target=run, args=([3, 0, 5, 1, 7, 2, 6, 4],))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not failures, failures[:5]


def test_second_context_gets_its_own_lds_opt_in(vnd):
    """A launch that needs more than 64 KiB of LDS must work on every context, not only on the
    first one that ran the kernel (the opt-in is per device; it used to be cached per kernel only).
    With one GPU both contexts sit on device 0; with two or more the second context is device 1."""
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    n_dev = _native.device_count()
    fir = np.zeros((30000, 2), np.float32)              # 0.6 s halo: the window passes 64 KiB even at the smallest tile
    rng = np.random.default_rng(2)
    for c in range(2):
        idx = np.sort(rng.choice(30000, 40, replace=False))
        fir[idx, c] = rng.choice([-1.0, 1.0], 40) * 0.5
    x = rng.uniform(-1, 1, (90000, 2)).astype(np.float32)
    want = O.convolve_velvet_noise(x, fir)
    arr = function_path_arrays(fir)
    for dev in ([0, 0] if n_dev < 2 else [0, 1, 0]):
        ctx = _native.Context(dev)
        table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
        text = table.describe(1, len(x), 2, vnd.MODE_EXACT)
        lds = int(text.split('lds=')[1].split('B')[0])
        assert lds > 65536, text
        assert np.array_equal(table.convolve_host(x, vnd.MODE_EXACT), want), f'device {dev}'
        got = table.convolve_host(x, vnd.MODE_FAST)
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))
        table.close()
        ctx.close()


def test_table_of_another_device_is_refused(vnd):
    from vndecorrelate_amd import _native
    if _native.device_count() < 2:
        pytest.skip('needs two GPUs')
    from vndecorrelate_amd.taps import function_path_arrays
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, sample_rate_hz=48000, seed=1)
    arr = function_path_arrays(fir)
    c0, c1 = _native.Context(0), _native.Context(1)
    table = _native.TapTable.create(c0, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    table.ctx = c1                                       # pretend it belongs to the other context
    with pytest.raises(ValueError):
        table.convolve_host(np.zeros((100, 2), np.float32))
    table.ctx = c0
    table.close()


def test_non_finite_weight_drops_tail_terms_like_the_reference(vnd):
    """out[:N-i] += x[i:] * w (decorrelation.py:656-658): a tap that reaches past the end of the
    signal contributes NOTHING there, even when w is inf (the LDS kernels' zero fill would give
    0 * inf = NaN).  Such tables take the index-testing kernel."""
    fir = np.zeros((64, 2), np.float32)
    fir[[1, 5, 40], 0] = [0.5, np.inf, -0.25]
    fir[[0, 17], 1] = [1.0, np.nan]
    x = np.random.default_rng(9).uniform(0.1, 1, (300, 2)).astype(np.float32)
    with np.errstate(all='ignore'):
        want = O.convolve_velvet_noise(x, fir)
    for mode in (vnd.MODE_EXACT, vnd.MODE_FMA, vnd.MODE_FAST):
        got = vnd.convolve_velvet_noise(x, fir, mode=mode)
        assert np.array_equal(np.isnan(got), np.isnan(want)), mode
        assert np.array_equal(np.isinf(got), np.isinf(want)), mode
        fin = np.isfinite(want)
        assert np.allclose(got[fin], want[fin], rtol=1e-6, atol=1e-6), mode
    assert np.isfinite(want[-4:, 0]).all() and np.isfinite(want[-10:, 1]).all()   # the dropped terms


def test_huge_tap_index_takes_the_gather_kernel(vnd):
    """Indices whose byte offsets would overflow the LDS kernels' 32-bit fields (2^29 and up) are
    legal for the reference; they must neither overflow nor fault (signed overflow in
    vnd_taps_create was undefined behaviour)."""
    from vndecorrelate_amd import _native
    ctx = _native.default_context()
    offs = np.array([0, 2, 3], np.int32)
    idx = np.array([3, (1 << 29) + 5, 0], np.int32)
    w = np.array([0.5, 2.0, -1.0], np.float32)
    table = _native.TapTable.create(ctx, offs, idx, w)
    assert 'conv_direct' in table.describe(1, 1000, 2, vnd.MODE_FAST)
    x = np.random.default_rng(1).uniform(-1, 1, (1000, 2)).astype(np.float32)
    got = table.convolve_host(x, vnd.MODE_EXACT)
    want = np.zeros_like(x)
    want[:-3, 0] = x[3:, 0] * np.float32(0.5)
    want[:, 1] = -x[:, 1]
    assert np.array_equal(got, want)
    table.close()


def test_first_paced_launch_of_a_fresh_context_inside_a_graph_capture(vnd):
    """A `*_dev` call only enqueues on the caller's stream (include/vnd_amd.h): the FIRST paced launch of a fresh context -
    cfg3's bench pool, two workgroups per CU pacing each other through the context's tile counters - must be legal inside a
    stream capture (the counters are allocated and zeroed by vnd_ctx_create, not by the launch, and never through the null
    stream), and the graph's replay must write what a plain launch writes."""
    import torch
    from oracle import c_oracle
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=128, num_outs=2, sample_rate_hz=48000,
                                                         log_distribution_strength=0.0, seed=1))
    batch, n = 24, 2880000
    ctx = _native.Context(0)
    table = _native.TapTable.create(ctx, arr.tap_offsets, arr.tap_index, arr.tap_weight)
    try:
        table.prepare(batch, n, 2, vnd.MODE_FAST)                  # the hipRTC build is host work: before the capture
        assert table.describe(batch, n, 2, vnd.MODE_FAST).startswith('conv_spec_window')
        x = torch.empty((batch, n, 2), dtype=torch.float32, device='cuda').uniform_(-1, 1)
        y = torch.zeros_like(x)
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            table.convolve_device(x.data_ptr(), y.data_ptr(), batch, n, 2, vnd.MODE_FAST, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        y.zero_()
        graph.replay()
        torch.cuda.synchronize()
        plain = torch.empty_like(y)
        table.convolve_device(x.data_ptr(), plain.data_ptr(), batch, n, 2, vnd.MODE_FAST, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(y, plain)
        want = c_oracle.convolve(x[batch - 1].cpu().numpy(), arr.tap_offsets, arr.tap_index, arr.tap_weight, threads=8)
        assert np.max(np.abs(y[batch - 1].cpu().numpy() - want)) <= 1e-6 * np.max(np.abs(want))
    finally:
        table.close()
        ctx.close()


def test_an_unregistered_tuning_name_is_an_error_code_not_an_abort(vnd):
    """Under VND_TUNING=1 (the test tier sets it) every tuning variable the library reads must be in its registry; a name
    that is not used to abort() the host process in debug builds - it is VND_ERR_INVALID now, and after this module's
    launches through every kernel family no such name has been recorded."""
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
    table = _native.TapTable.create(_native.default_context(), arr.tap_offsets, arr.tap_index, arr.tap_weight)
    text = table.describe(64, 480000, 2, vnd.MODE_FAST)                 # plans a launch: reads the geometry variables
    assert text.startswith('conv_spec'), text
    table.close()
