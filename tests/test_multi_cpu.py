"""CPU tier: the one-process, several-GPU form of the batched mode (vndecorrelate_amd/multi.py) - how a batch is cut over
the devices and put back together, with a checker injected in place of the GPU workers (the product has no CPU path)."""
import threading

import numpy as np
import pytest

from oracle import vnd_oracle as O
from vndecorrelate_amd import multi
from vndecorrelate_amd.taps import TapArrays, function_path_arrays


def _arrays():
    fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    return fir, function_path_arrays(fir)


class OracleWorker:
    """Stands where a GPU would: the oracle on the block it is handed, and a note of who ran what."""

    def __init__(self, device, arrays: TapArrays, log):
        self.device, self.log = device, log
        self.fir = np.zeros((int(arrays.tap_index.max()) + 1, arrays.num_channels), np.float32)
        for c in range(arrays.num_channels):
            lo, hi = arrays.tap_offsets[c], arrays.tap_offsets[c + 1]
            self.fir[arrays.tap_index[lo:hi], c] = arrays.tap_weight[lo:hi]

    def convolve(self, x, out, mode):
        assert out.flags.c_contiguous and x.flags.c_contiguous and out.shape[0] == x.shape[0]
        self.log.append((self.device, x.shape[0], threading.current_thread().name))
        for b in range(x.shape[0]):
            out[b] = O.convolve_velvet_noise(x[b], self.fir)

    def decorrelate(self, x, out, mode, *, ms_encode, width, normalize):
        self.log.append((self.device, x.shape[0], 'decorrelate', ms_encode, width, normalize))
        for b in range(x.shape[0]):
            out[b] = O.convolve_velvet_noise(x[b], self.fir) * 0.5

    def close(self):
        self.log.append((self.device, 'closed'))


def _pool(n_devices, log, **kw):
    made = []

    def factory(devices, arrays):
        made.append(len(devices))
        return [OracleWorker(d, arrays, log) for d in devices]
    pool = multi.DevicePool(list(range(n_devices)), worker_factory=factory, **kw)
    pool.made = made
    return pool


@pytest.mark.parametrize('n_devices', [1, 2, 3, 8])
@pytest.mark.parametrize('streams', [1, 5, 8, 13])
def test_blocks_are_contiguous_ragged_and_reassembled(n_devices, streams):
    fir, arrays = _arrays()
    x = np.random.default_rng(streams).uniform(-1, 1, (streams, 3000, 2)).astype(np.float32)
    want = np.stack([O.convolve_velvet_noise(x[b], fir) for b in range(streams)])
    log = []
    pool = _pool(n_devices, log)
    out = np.full_like(x, np.nan)
    got = pool.map_streams(arrays, x, out, 'convolve', 0)
    assert got is out and np.array_equal(out, want)
    # vnd_shard_range's cut: contiguous, in device order, the remainder one each to the first devices, nothing for a device without streams
    base, extra = divmod(streams, n_devices)
    counts = [base + (1 if d < extra else 0) for d in range(n_devices)]
    assert pool.last_blocks == [(sum(counts[:d]), counts[d]) for d in range(n_devices)]
    assert sorted((d, c) for d, c, _ in log) == [(d, c) for d, c in enumerate(counts) if c]
    if sum(1 for c in counts if c) > 1:
        assert all(name.startswith('vnd-dev') for _, _, name in log)         # one host thread per busy device
    pool.close()


def test_the_table_is_replicated_once_per_content_and_evicted_lru():
    fir, arrays = _arrays()
    log = []
    pool = _pool(2, log, cache_tables=2)
    x = np.zeros((2, 2000, 2), np.float32)
    for _ in range(3):
        pool.map_streams(arrays, x, np.empty_like(x), 'convolve', 0)
    assert pool.made == [2]                                                  # built once, used three times
    others = []
    for seed in (2, 3):
        others.append(function_path_arrays(O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=seed)))
        pool.map_streams(others[-1], x, np.empty_like(x), 'convolve', 0)
    assert pool.made == [2, 2, 2]
    pool.map_streams(arrays, x, np.empty_like(x), 'convolve', 0)             # the first table was evicted (capacity 2): built again
    assert pool.made == [2, 2, 2, 2]
    pool.close()
    assert [e for e in log if e[1:] == ('closed',)]


def test_a_failing_device_fails_the_call_after_every_block_has_finished():
    fir, arrays = _arrays()
    done = []

    class Worker:
        def __init__(self, d):
            self.d = d

        def convolve(self, x, out, mode):
            if self.d == 1:
                raise RuntimeError('device 1 fell over')
            out[:] = 1.0
            done.append(self.d)

        def close(self):
            pass
    pool = multi.DevicePool([0, 1, 2], worker_factory=lambda devices, a: [Worker(d) for d in devices])
    x = np.zeros((6, 100, 2), np.float32)
    with pytest.raises(RuntimeError, match='device 1 fell over'):
        pool.map_streams(arrays, x, np.zeros_like(x), 'convolve', 0)
    assert sorted(done) == [0, 2]
    pool.close()


def test_decorrelate_blocks_carry_the_stage_arguments():
    fir, arrays = _arrays()
    log = []
    pool = _pool(3, log)
    x = np.random.default_rng(0).uniform(-1, 1, (4, 2500, 1)).astype(np.float32)       # mono in, stereo out: fan-out blocks
    out = np.empty((4, 2500, 2), np.float32)
    with pytest.raises(Exception):
        pool.map_streams(arrays, x, np.empty((3, 2500, 2), np.float32), 'decorrelate', 0, ms_encode=True, width=None, normalize=1)
    xs = np.repeat(x, 2, axis=2)
    pool.map_streams(arrays, xs, out, 'decorrelate', 0, ms_encode=True, width=0.5, normalize=1)
    assert sorted(e[:2] for e in log) == [(0, 2), (1, 1), (2, 1)] and all(e[3:] == (True, 0.5, 1) for e in log)
    assert np.array_equal(out[3], O.convolve_velvet_noise(xs[3], fir) * 0.5)
    pool.close()


def test_device_lists_are_validated():
    assert multi.resolve_devices('all', available=4) == [0, 1, 2, 3]
    assert multi.resolve_devices([2, 0], available=4) == [2, 0]
    for bad in ([], [0, 0], [4], [-1], 'some'):
        with pytest.raises(ValueError):
            multi.resolve_devices(bad, available=4)
    with pytest.raises(RuntimeError):
        multi.resolve_devices('all', available=0)
    assert multi.blocks(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    with pytest.raises(ValueError):
        multi.DevicePool([0], table_transport='pigeon')


def test_the_gpu_workers_fail_loudly_without_a_device():
    """No CPU path: the default factory needs real devices (here: none), and says so."""
    from vndecorrelate_amd import _native
    if _native.device_count() > 0:
        pytest.skip('a GPU is visible')
    fir, arrays = _arrays()
    pool = multi.DevicePool([0])
    x = np.zeros((2, 100, 2), np.float32)
    with pytest.raises((RuntimeError, ValueError)):
        pool.map_streams(arrays, x, np.empty_like(x), 'convolve', 0)
    pool.close()
    import vndecorrelate_amd.decorrelation as vnd
    with pytest.raises(RuntimeError):
        vnd.convolve_velvet_noise_batched(x, fir, devices='all')


def test_two_callers_share_a_pool():
    """Two host threads through the same pool at once (the GPU workers serialise per device on their context's mutex; here the
    checker has no such lock): every caller gets its own batch back, whole."""
    fir, arrays = _arrays()
    log = []
    pool = _pool(3, log)
    rng = np.random.default_rng(5)
    batches = [rng.uniform(-1, 1, (7, 1500, 2)).astype(np.float32), rng.uniform(-1, 1, (4, 2100, 2)).astype(np.float32)]
    outs = [np.empty_like(b) for b in batches]
    errors = []

    def caller(k):
        try:
            for _ in range(5):
                pool.map_streams(arrays, batches[k], outs[k], 'convolve', 0)
        except Exception as exc:                        # pragma: no cover
            errors.append(exc)
    threads = [threading.Thread(target=caller, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for k in range(2):
        want = np.stack([O.convolve_velvet_noise(batches[k][b], fir) for b in range(len(batches[k]))])
        assert np.array_equal(outs[k], want)
    assert pool.made == [3]
    pool.close()


def test_a_failed_rccl_communicator_means_an_upload_per_device_not_a_failed_call(monkeypatch):
    """The default table transport of the several-device call is one RCCL broadcast; where the single-process communicator cannot be
    made (no librccl, ncclCommInitAll refusing the fabric) every device deserialises the 8-bytes-per-tap image itself - still the
    GPUs' tables, a warning, and `last_transport` says which way it went.  (Fake contexts and tables: no device here.)"""
    from vndecorrelate_amd import _native
    _, arrays = _arrays()
    made = []

    class FakeTable:
        def __init__(self, ctx, image=None):
            self.ctx, self.image = ctx, image

        @classmethod
        def create(cls, ctx, offsets, index, weight, **kw):
            return cls(ctx, arrays.to_bytes())

        @classmethod
        def from_bytes(cls, ctx, image):
            made.append((ctx, len(image)))
            return cls(ctx, image)

        def to_bytes(self):
            return self.image

    class BrokenRccl:
        def __init__(self):
            raise RuntimeError('librccl.so could not be loaded')
    monkeypatch.setattr(_native, 'context_for', lambda d: f'ctx{d}')
    monkeypatch.setattr(_native, 'TapTable', FakeTable)
    monkeypatch.setattr(multi, '_Rccl', BrokenRccl)
    pool = multi.DevicePool([0, 1, 2])
    with pytest.warns(UserWarning, match='uploading the table to each device'):
        workers = pool._gpu_workers(pool.devices, arrays)
    assert [w.ctx for w in workers] == ['ctx0', 'ctx1', 'ctx2'] and [c for c, _ in made] == ['ctx1', 'ctx2']
    assert pool.last_transport.startswith('upload (rccl unavailable') and all(w.table.image == arrays.to_bytes() for w in workers)
    pool._threads.shutdown(wait=True)
