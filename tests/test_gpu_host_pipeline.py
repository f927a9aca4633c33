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
