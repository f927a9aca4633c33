"""GPU tier: the synchronous host API's pipelines.  A few long streams are cut in TIME (upload || kernel || download
over pieces of one stream, csrc/vnd_amd.hip: host_time_pipeline): the seams must not show - bit-identical to the
oracle in exact mode, whatever the length, channel count or batch."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import vnd_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [300007, 480000, 8 * 4096, 8 * 4096 + 1, 1234567])
@pytest.mark.parametrize('batch,cx,pieces', [(1, 2, 5), (3, 2, 2), (1, 1, 8), (1, 2, 3)])
def test_time_pieces_of_a_long_stream_are_seamless(golden, monkeypatch, n, batch, cx, pieces):
    """n is not a multiple of the 4096-frame piece marks; pieces end inside the filter's reach of each other; the last
    piece is the stream's tail.  Exact mode: the oracle's bytes.  Fast mode: its tolerance.  (The library itself cuts
    only page-locked streams of 16 MB and more - every extra copy call costs ~50 us here - so the piece count is forced.)"""
    import vndecorrelate_amd.decorrelation as d
    monkeypatch.setenv('VND_HOST_TIME_PIECES', str(pieces))
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    fir = golden.fir('g48k_k30')
    a = function_path_arrays(fir)
    table = _native.TapTable.create(_native.default_context(), a.tap_offsets, a.tap_index, a.tap_weight)
    rng = np.random.default_rng(n + batch)
    x = rng.uniform(-1, 1, (batch, n, cx)).astype(np.float32)
    offs, idx, w = O.fir_to_taps(fir)
    got = table.convolve_host(x, d.MODE_EXACT)
    fast = table.convolve_host(x, d.MODE_FAST)
    assert got.shape == (batch, n, 2)
    for b in range(batch):
        xb = x[b] if cx == 2 else np.repeat(x[b], 2, axis=1)
        want = c_oracle.convolve(np.ascontiguousarray(xb), offs, idx, w)
        assert np.array_equal(got[b], want), (n, batch, cx, b, int(np.argmax(np.any(got[b] != want, axis=1))))
        assert np.max(np.abs(fast[b].astype(np.float64) - want)) <= 1e-6 * np.max(np.abs(want))
    table.close()


def test_time_pieces_only_for_a_few_long_streams(golden):
    """Short signals and large batches keep their one-piece / whole-stream pipelines (same results either way: this
    guards the thresholds, through the public function)."""
    import vndecorrelate_amd.decorrelation as d
    fir = golden.fir('g48k_k30')
    rng = np.random.default_rng(3)
    for shape in ((2000, 2), (40000, 2), (6, 70000, 2)):
        x = rng.uniform(-1, 1, shape).astype(np.float32)
        y = d.convolve_velvet_noise_batched(x, fir) if x.ndim == 3 else d.convolve_velvet_noise(x, fir)
        ref = np.stack([O.convolve_velvet_noise(s, fir) for s in x]) if x.ndim == 3 else O.convolve_velvet_noise(x, fir)
        assert np.array_equal(y, ref), shape


@pytest.mark.parametrize('shape', [(480000, 2), (3, 300007, 2), (2, 131072, 1), (16, 48000, 2), (250001, 2)])
def test_page_locked_buffers_are_convolved_in_place(golden, monkeypatch, shape):
    """Both buffers page-locked (the input from the library's own pool here; torch's pin_memory is the same kind of
    memory): no staging - the kernel reads x and writes y across PCIe.  The library says so (vnd_host_buffers_mapped), the
    exact result is the oracle's bytes and equals the staged path's, the fast result keeps its tolerance.  Below the
    pool's 1 MiB threshold the result is an ordinary array and the call is staged."""
    import vndecorrelate_amd.decorrelation as d
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    fir = golden.fir('g48k_k30')
    a = function_path_arrays(fir)
    table = _native.TapTable.create(_native.default_context(), a.tap_offsets, a.tap_index, a.tap_weight)
    rng = np.random.default_rng(len(shape) * 1000 + shape[-2])
    pageable = rng.uniform(-1, 1, shape).astype(np.float32)
    x = _native.pinned_pool.empty(shape, np.float32)
    x[...] = pageable
    out_shape = shape[:-1] + (2,)
    y_probe = _native.pinned_pool.empty(out_shape, np.float32)
    assert _native.host_buffers_mapped(x, y_probe)
    assert not _native.host_buffers_mapped(pageable, y_probe) and not _native.host_buffers_mapped(x, np.empty(out_shape, np.float32))
    assert _native.host_buffers_mapped(x[..., 1000:, :], y_probe[..., 1000:, :]) if x.ndim == 2 else True      # a view: still mapped
    del y_probe
    offs, idx, w = O.fir_to_taps(fir)
    got, fast = table.convolve_host(x, d.MODE_EXACT), table.convolve_host(x, d.MODE_FAST)
    monkeypatch.setenv('VND_HOST_DIRECT', '0')                    # (read live: tests run with VND_TUNING=1)
    staged = table.convolve_host(x, d.MODE_EXACT)
    monkeypatch.delenv('VND_HOST_DIRECT')
    assert np.array_equal(got, staged)
    xs = x if x.ndim == 3 else x[None]
    for b in range(xs.shape[0]):
        xb = xs[b] if xs.shape[-1] == 2 else np.repeat(xs[b], 2, axis=1)
        want = c_oracle.convolve(np.ascontiguousarray(xb), offs, idx, w)
        gb, fb = (got[b], fast[b]) if x.ndim == 3 else (got, fast)
        assert np.array_equal(gb, want), (shape, b)
        assert np.max(np.abs(fb.astype(np.float64) - want)) <= 1e-6 * np.max(np.abs(want))
    # the public function on a page-locked input: same bytes
    if x.ndim == 2 and shape[-1] == 2:
        assert np.array_equal(d.convolve_velvet_noise(x, fir), got)
        # a view that starts 3 frames in (8-byte aligned only, still page-locked): in place through the kernels that take any alignment
        view = x[3:]
        assert view.flags.c_contiguous and view.ctypes.data % 16 == 8
        want_view = c_oracle.convolve(np.ascontiguousarray(view), offs, idx, w)
        assert np.array_equal(table.convolve_host(view, d.MODE_EXACT), want_view)
        fast_view = table.convolve_host(view, d.MODE_FAST)
        assert np.max(np.abs(fast_view.astype(np.float64) - want_view)) <= 1e-6 * np.max(np.abs(want_view))
    table.close()
