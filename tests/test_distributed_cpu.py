"""CPU tier: the N>1 path with world_size-2 gloo.  The sharding and the
tap-table broadcast are the product code; the per-rank compute is injected (the
oracle, as checker) because this box has no GPU."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from vndecorrelate_amd.distributed import shard_range
from vndecorrelate_amd.taps import TapArrays, class_path_arrays, function_path_arrays


def test_shard_range_partitions_everything():
    for total in (0, 1, 7, 8, 1024, 1027):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_table_image_roundtrip(golden):
    fn = function_path_arrays(golden.fir('g96k_k64_c8'))
    back = TapArrays.from_bytes(fn.to_bytes())
    assert np.array_equal(back.tap_index, fn.tap_index) and np.array_equal(back.tap_weight, fn.tap_weight)
    assert back.seg_offsets is None and back.chan_flags is None and len(fn.to_bytes()) == 32 + 4 * 9 + 8 * 512
    cls = class_path_arrays(golden.class_taps('v44k_ch0_lr', 2), (0.85, 0.55, 0.35, 0.2), True)
    back = TapArrays.from_bytes(cls.to_bytes())
    for name in ('tap_offsets', 'tap_index', 'tap_weight', 'seg_offsets', 'seg_end', 'seg_gain', 'chan_flags'):
        assert np.array_equal(getattr(back, name), getattr(cls, name)), name
    assert back.apply_gain is True
    with pytest.raises(ValueError):
        TapArrays.from_bytes(b'\0' * 64)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        import pathlib
        sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
        from oracle import c_oracle
        from oracle import vnd_oracle as O
        from vndecorrelate_amd.distributed import ShardedDecorrelator

        def checker_backend(arrays):
            def run(x_local, mode):
                return c_oracle.convolve(x_local, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight)
            return run

        fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, sample_rate_hz=48000, seed=1)
        arrays = function_path_arrays(fir) if rank == 0 else None      # only the source rank has the table
        sharded = ShardedDecorrelator(arrays, src=0, backend=checker_backend)
        assert sharded.world_size == world and sharded.rank == rank
        assert sharded.image == function_path_arrays(fir).to_bytes()   # every rank got rank 0's bytes
        x = np.random.default_rng(0).uniform(-1, 1, (7, 3000, 2)).astype(np.float32)   # ragged: 4 + 3
        y_local = sharded.convolve_global(x)
        start, count = sharded.shard(7)
        assert y_local.shape == (count, 3000, 2)
        np.save(os.path.join(out_dir, f'y{rank}.npy'), y_local)
        np.save(os.path.join(out_dir, f'span{rank}.npy'), np.array([start, count]))

        # the optimiser's scan sharded over candidates: 5 candidates -> 3 + 2, scores all-gathered
        from vndecorrelate_amd.distributed import sharded_grid_scan
        kw = dict(angle_limit=float(np.pi / 4), lambda_mean=5.0, lambda_skew=2.0, lambda_correlation=15.0,
                  lambda_penalty=1e3)
        kappas = [0.0, 0.25, 0.5, 0.75, 1.0]
        sig = np.random.default_rng(3).uniform(-1, 1, (4000, 2)).astype(np.float32)

        def checker_scorer(signal, candidates, **objective):          # the oracle's objective per candidate
            return np.array([O.symmetry_aware_objective(
                O.decorrelate(signal.copy(), sample_rate_hz=48000, log_distribution_strength=k,
                              filtered_channels=(0,), mode='LR', normalize=False, seed=1), **objective)
                for k in candidates])

        scores = sharded_grid_scan(sig, kappas, scorer=checker_scorer, **kw)
        assert scores.shape == (5,) and np.array_equal(scores, checker_scorer(sig, kappas, **kw))
        assert sharded_grid_scan(sig, [], scorer=checker_scorer, **kw).shape == (0,)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_run(tmp_path):
    from oracle import c_oracle
    from oracle import vnd_oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    spans = [np.load(tmp_path / f'span{r}.npy').tolist() for r in range(world)]
    assert spans == [[0, 4], [4, 3]]
    y = np.concatenate([np.load(tmp_path / f'y{r}.npy') for r in range(world)])
    fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, sample_rate_hz=48000, seed=1)
    x = np.random.default_rng(0).uniform(-1, 1, (7, 3000, 2)).astype(np.float32)
    offs, idx, w = O.fir_to_taps(fir)
    assert np.array_equal(y, c_oracle.convolve(x, offs, idx, w))
    for b in range(7):                                                  # streams are independent
        assert np.array_equal(y[b], O.convolve_velvet_noise(x[b], fir))


def _time_shard_worker(rank, world, port, out_dir, cuts):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        import pathlib
        sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
        from oracle import c_oracle
        from oracle import vnd_oracle as O
        from vndecorrelate_amd.distributed import ShardedDecorrelator

        def checker_backend(arrays):
            return lambda x_local, mode: c_oracle.convolve(x_local, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight)

        fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, sample_rate_hz=48000, seed=1)
        sharded = ShardedDecorrelator(function_path_arrays(fir) if rank == 0 else None, src=0, backend=checker_backend)
        x = np.random.default_rng(5).uniform(-1, 1, (cuts[-1], 2)).astype(np.float32)
        y_local = sharded.convolve_time_shard(x[cuts[rank]:cuts[rank + 1]])
        assert y_local.shape == (cuts[rank + 1] - cuts[rank], 2)
        np.save(os.path.join(out_dir, f't{rank}.npy'), y_local)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('cuts', [
    (0, 9000, 20000),                 # two ranks, slices longer than the 1260-frame halo
    (0, 5000, 5400, 5401, 12000),     # four ranks: the halo of rank 0 spans ranks 1, 2 and 3
    (0, 4000, 4000, 7000),            # an empty slice in the middle
    (0, 6000, 6800),                  # the last slice is shorter than the halo: the stream's true end
])
def test_one_stream_cut_in_time_over_ranks(tmp_path, cuts):
    """A long stream cut over the ranks with one forward-halo exchange equals the unsharded call, bit for bit."""
    from oracle import vnd_oracle as O
    world = len(cuts) - 1
    mp.spawn(_time_shard_worker, args=(world, _free_port(), str(tmp_path), cuts), nprocs=world, join=True)
    y = np.concatenate([np.load(tmp_path / f't{r}.npy') for r in range(world)])
    fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, sample_rate_hz=48000, seed=1)
    x = np.random.default_rng(5).uniform(-1, 1, (cuts[-1], 2)).astype(np.float32)
    assert np.array_equal(y, O.convolve_velvet_noise(x, fir))


def test_c_abi_shard_range_matches_the_python_one():
    """vnd_shard_range (for hosts that shard without torch.distributed) cuts what shard_range cuts."""
    from vndecorrelate_amd import _native
    for total in (0, 1, 7, 8, 1024, 1027, 70000):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert _native.shard_range(total, world, rank) == shard_range(total, world, rank)
    for bad in ((4, 2, 2), (4, 0, 0), (-1, 2, 0), (4, 2, -1)):
        with pytest.raises(ValueError):                    # VND_ERR_INVALID, as the Python shard_range raises
            _native.shard_range(*bad)
