"""GPU tier: the many-stream mode with more than one rank on real hardware, and cfg4 at full size.

Two ranks, one process each (torch.multiprocessing): with two or more GPUs they use `nccl` (RCCL over
xGMI) and one device each - the production configuration; on a one-GPU box both ranks share device 0
and broadcast over `gloo`, which still drives the product code end to end: table built on rank 0 only,
broadcast, contiguous shards, device-resident sharded convolution, no data-path collective."""
import hashlib
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, backend, n_devices, result_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    device_index = rank % n_devices
    os.environ['VND_DEVICE'] = str(device_index)
    import sys
    import pathlib
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(device_index)
    device = torch.device('cuda', device_index)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import vndecorrelate_amd.decorrelation as vnd
        from oracle import vnd_oracle as O
        from vndecorrelate_amd.distributed import ShardedDecorrelator
        from vndecorrelate_amd.taps import function_path_arrays
        arrays = None
        fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
        if rank == 0:                                   # only the source rank builds the table
            arrays = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2,
                                                                    sample_rate_hz=48000, seed=1))
        sharded = ShardedDecorrelator(arrays, src=0, device=device if backend == 'nccl' else None)
        streams, n = 7, 30000                            # ragged: 4 + 3
        x_all = np.random.default_rng(17).uniform(-1, 1, (streams, n, 2)).astype(np.float32)
        start, count = sharded.shard(streams)
        x_local = torch.from_numpy(x_all[start:start + count]).to(device)
        y_exact = sharded.convolve_local_device(x_local, mode=vnd.MODE_EXACT)
        y_fast = sharded.convolve_local_device(x_local, mode=vnd.MODE_FAST)
        torch.cuda.synchronize()
        ok = True
        for i in range(count):
            want = O.convolve_velvet_noise(x_all[start + i], fir)
            ok &= bool(np.array_equal(y_exact[i].cpu().numpy(), want))
            ok &= bool(np.max(np.abs(y_fast[i].cpu().numpy() - want)) <= 1e-6 * np.max(np.abs(want)))
        # host-array form agrees
        ok &= bool(np.array_equal(sharded.convolve_local(x_all[start:start + count], vnd.MODE_EXACT), y_exact.cpu().numpy()))
        # ONE long stream cut in time over the ranks (forward-halo send/recv), GPU kernels per rank:
        # rank 0 takes 50 000 frames, rank 1 the last 900 - shorter than the 1260-frame halo
        long_x = np.random.default_rng(23).uniform(-1, 1, (50900, 2)).astype(np.float32)
        cut = (0, 50000, 50900)
        y_time = sharded.convolve_time_shard(long_x[cut[rank]:cut[rank + 1]], vnd.MODE_EXACT,
                                             device=device if backend == 'nccl' else None)
        ok &= bool(np.array_equal(y_time, O.convolve_velvet_noise(long_x, fir)[cut[rank]:cut[rank + 1]]))
        with open(os.path.join(result_dir, f'rank{rank}.txt'), 'w') as f:
            f.write(f'{int(ok)} {start} {count} {backend} {device_index}')
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_on_the_gpu(tmp_path):
    import torch
    import torch.multiprocessing as mp
    n_devices = torch.cuda.device_count()
    if n_devices < 1:
        pytest.skip('needs a GPU')
    backend = 'nccl' if n_devices >= 2 else 'gloo'
    mp.spawn(_rank_main, args=(2, _free_port(), backend, n_devices, str(tmp_path)), nprocs=2, join=True)
    spans = []
    for r in range(2):
        ok, start, count, used, dev = (tmp_path / f'rank{r}.txt').read_text().split()
        assert ok == '1', f'rank {r} disagrees with the oracle ({used}, device {dev})'
        spans.append((int(start), int(count)))
    assert spans == [(0, 4), (4, 3)]


def test_cfg4_full_size_is_bit_exact_per_stream():
    """BASELINE configs[3] whole: 1024 independent 1 s stereo streams in one launch, exact mode against
    the C oracle (sha256 per stream), fast mode within 1e-6 of peak, through the host batch API."""
    import vndecorrelate_amd.decorrelation as vnd
    from oracle import c_oracle
    from oracle import vnd_oracle as O
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    x = np.random.default_rng(4).uniform(-1, 1, (1024, 48000, 2)).astype(np.float32)
    offs, idx, w = O.fir_to_taps(fir)
    want = c_oracle.convolve(x, offs, idx, w, threads=min(os.cpu_count() or 1, 16))
    got = vnd.convolve_velvet_noise_batched(x, fir, mode=vnd.MODE_EXACT)
    for b in range(0, 1024):
        if hashlib.sha256(got[b].tobytes()).digest() != hashlib.sha256(want[b].tobytes()).digest():
            raise AssertionError(f'stream {b} differs from the oracle')
    fast = vnd.convolve_velvet_noise_batched(x, fir, mode=vnd.MODE_FAST)
    assert np.max(np.abs(fast - want)) <= 1e-6 * np.max(np.abs(want))
    # the pipelined host path returns page-locked result arrays; they behave like any other ndarray
    assert fast.flags.writeable and fast.dtype == np.float32 and fast.shape == x.shape
    del got, fast


def _rccl():
    import ctypes
    for name in ('librccl.so.1', 'librccl.so'):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    return None


def test_table_broadcast_through_the_c_abi_over_rccl():
    """vnd_taps_broadcast_rccl on a communicator this test creates straight from librccl (no torch):
    with one GPU a communicator of one rank - the root keeps its table, the call drives RCCL and returns;
    with two or more GPUs see test_two_rank_rccl_broadcast."""
    import ctypes
    import vndecorrelate_amd.decorrelation as vnd
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    rccl = _rccl()
    if rccl is None:
        pytest.skip('librccl.so not found')

    class UniqueId(ctypes.Structure):
        _fields_ = [('internal', ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        ctx = _native.default_context()
        arrays = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2,
                                                                sample_rate_hz=48000, seed=1))
        table = _native.TapTable.create(ctx, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight)
        same = _native.TapTable.broadcast_rccl(ctx, table, root=0, rank=0, comm=comm.value)
        assert same is table and same.to_bytes() == arrays.to_bytes()
        with pytest.raises(ValueError):                    # the root must have a table to send
            _native.TapTable.broadcast_rccl(ctx, None, root=0, rank=0, comm=comm.value)
        table.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def _rccl_rank(rank, world, uid_bytes, result_dir):
    import ctypes
    os.environ['VND_DEVICE'] = str(rank)
    import sys
    import pathlib
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
    import torch
    torch.cuda.set_device(rank)
    import vndecorrelate_amd.decorrelation as vnd
    from vndecorrelate_amd import _native
    from vndecorrelate_amd.taps import function_path_arrays
    rccl = _rccl()

    class UniqueId(ctypes.Structure):
        _fields_ = [('internal', ctypes.c_char * 128)]

    uid = UniqueId.from_buffer_copy(uid_bytes)
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
    ctx = _native.default_context()
    arrays = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2,
                                                            sample_rate_hz=48000, seed=1))
    table = _native.TapTable.create(ctx, arrays.tap_offsets, arrays.tap_index, arrays.tap_weight) if rank == 0 else None
    got = _native.TapTable.broadcast_rccl(ctx, table, root=0, rank=rank, comm=comm.value)
    ok = got.to_bytes() == arrays.to_bytes()
    first, count = _native.shard_range(7, world, rank)
    x = np.random.default_rng(17).uniform(-1, 1, (7, 20000, 2)).astype(np.float32)[first:first + count]
    from oracle import vnd_oracle as O
    fir = O.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    y = got.convolve_host(x, vnd.MODE_EXACT)
    ok &= all(np.array_equal(y[i], O.convolve_velvet_noise(x[i], fir)) for i in range(count))
    with open(os.path.join(result_dir, f'rccl{rank}.txt'), 'w') as f:
        f.write(str(int(ok)))
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    rccl.ncclCommDestroy(comm)


def test_two_rank_rccl_broadcast(tmp_path):
    """Two GPUs, two processes, no torch.distributed: RCCL communicator from librccl, table built on rank 0
    only, vnd_taps_broadcast_rccl, vnd_shard_range, exact convolution per shard."""
    import ctypes
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL does not put two ranks on one device)')
    rccl = _rccl()
    if rccl is None:
        pytest.skip('librccl.so not found')
    uid = (ctypes.c_char * 128)()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    mp.spawn(_rccl_rank, args=(2, bytes(uid), str(tmp_path)), nprocs=2, join=True)
    assert [(tmp_path / f'rccl{r}.txt').read_text() for r in range(2)] == ['1', '1']


@pytest.mark.parametrize('how', ['plain', 'torchrun'])
def test_bench_runs_with_two_ranks(tmp_path, how):
    """bench.py's N > 1 plumbing both ways it is started: `plain` = `python bench.py --gpus 2 ...` (the parent starts its
    own ranks as a child `python -m torch.distributed.run` and relays rank 0's line; it never touches the GPU), `torchrun` =
    that command given directly.  RANK / LOCAL_RANK / WORLD_SIZE from the environment, table broadcast from rank 0,
    barriers around the timed region, max over ranks, ONE JSON line from rank 0 with the cfg4 strong cut over the ranks
    under `config.cfg4_strong` and the rank count the process group's own all-reduce saw.
    With two GPUs: RCCL (`nccl`), one device per rank.  On a one-GPU box both ranks share device 0 and the process group
    is `gloo` - everything but the RCCL transport itself (DESIGN.md 6 lists the lines that have therefore never run
    with more than one rank)."""
    import json
    import pathlib
    import subprocess
    import sys
    import torch
    repo = pathlib.Path(__file__).resolve().parents[1]
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for name in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(name, None)
    if not two:
        env['VND_BENCH_FORCE_DEVICE'] = '0'
    detail = tmp_path / 'detail.json'
    args = ['--gpus', '2', '--steps', '3', '--warmup', '1', '--pool', '16', '--min-warmup-ms', '10', '--backend', 'nccl' if two else 'gloo',
            '--detail', str(detail)]
    if how == 'plain':
        cmd = [sys.executable, str(repo / 'bench.py'), *args]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), str(repo / 'bench.py'), *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(repo))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    if how == 'torchrun':                                           # (gloo prints its connection notes on the ranks' stdout; the plain form relays those to stderr)
        lines = [l for l in lines if l.startswith('{"metric"')]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), r.stdout[-2000:]     # rank 0 alone reports, and nothing else is on stdout
    assert len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['scaling'] == 'weak' and line['value'] > 0
    config = line['config']
    assert config['parity'] <= 1e-6 and config['world_size'] == 2 and config['ranks_seen'] == 2
    assert config['backend'] == ('nccl' if two else 'gloo')
    strong = config['cfg4_strong']
    assert strong['ranks'] == 2 and strong['streams_on_rank0'] == 512 and strong['ms_per_pass_max_over_ranks'] > 0
    assert 'cpu_baseline' not in line and 'cfg3_frac' not in config   # N = 1 only
    full = json.loads(detail.read_text())
    assert full['cfg4_strong']['parity_vs_oracle_of_peak'] <= 1e-6 and 'secondary' not in full


def test_bench_rehearsal_with_four_ranks(tmp_path):
    """The SCALE step's command shape (`python bench.py --gpus N`, self-launched ranks) at the largest N one lease safely allows.  The
    pool this suite runs on admits at most SIX processes on a card at once - this test runner is one of them (a run with six ranks
    was ended by the box's process guard, round 6) - so the N = 8 case itself cannot be rehearsed on a one-GPU box; four ranks (all
    on device 0 over `gloo`, or one per device over RCCL where four GPUs are visible) exercise what grows with N: concurrent
    hipRTC builds of one table on one cold disk cache, the relay of every rank's stdout, the cfg4 cut (1024 streams over 4 ranks),
    barriers and the max-over-ranks all-reduce - and the wall time, which is recorded."""
    import json
    import pathlib
    import subprocess
    import sys
    import time
    import torch
    repo = pathlib.Path(__file__).resolve().parents[1]
    ranks = 4
    real = torch.cuda.device_count() >= ranks
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', VND_SPEC_CACHE_DIR=str(tmp_path / 'cache'))      # a cold code-object cache: every rank meets the table first
    for name in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(name, None)
    if not real:
        env['VND_BENCH_FORCE_DEVICE'] = '0'
    detail = tmp_path / 'detail.json'
    cmd = [sys.executable, str(repo / 'bench.py'), '--gpus', str(ranks), '--steps', '3', '--warmup', '1', '--pool', '64', '--min-warmup-ms', '10',
           '--backend', 'nccl' if real else 'gloo', '--detail', str(detail)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(repo))
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), r.stdout[-2000:]
    line = json.loads(lines[0])
    config = line['config']
    assert line['n_gpus'] == ranks and config['world_size'] == ranks and config['ranks_seen'] == ranks and config['parity'] <= 1e-6
    strong = config['cfg4_strong']
    assert strong['ranks'] == ranks and strong['streams_on_rank0'] == 256 and strong['Msamples_s'] > 0
    assert list(config)[:12].count('cfg4_strong') == 1               # among the keys the driver's record keeps
    print(f'bench.py --gpus {ranks} --pool 64 on {"four devices" if real else "one device (gloo)"}: {wall:.1f} s wall, cold kernel cache')
    (repo / 'gpurun_out').mkdir(exist_ok=True)
    (repo / 'gpurun_out' / 'bench_four_ranks.json').write_text(json.dumps({'wall_s': round(wall, 1), 'devices': ranks if real else 1, 'line': line}))


def test_one_process_device_lists_through_the_drop_in_calls():
    """`convolve_velvet_noise_batched(x, fir, devices=...)` and `VelvetNoise.decorrelate_batched(x, devices=...)`: the batch cut
    over the listed devices from THIS process (multi.DevicePool - one context and one host thread per device, blocks written
    straight into one result).  With one visible device the list is [0] and the result must be the unsharded call's, bit for
    bit, in both modes; with more, 'all' spreads a ragged batch over every device (table replicated over a single-process RCCL
    communicator) and must still equal it."""
    import vndecorrelate_amd.decorrelation as vnd
    from vndecorrelate_amd import _native, multi
    n_dev = _native.device_count()
    fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
    x = np.random.default_rng(41).uniform(-1, 1, (11, 30000, 2)).astype(np.float32)
    try:
        for mode in (vnd.MODE_EXACT, vnd.MODE_FAST):
            want = vnd.convolve_velvet_noise_batched(x, fir, mode=mode)
            for devices in ([0], 'all'):
                got = vnd.convolve_velvet_noise_batched(x, fir, mode=mode, devices=devices)
                assert got.dtype == np.float32 and np.array_equal(got, want), (mode, devices)
        pool = multi.pool_for('all')
        assert [c for _, c in pool.last_blocks] == [11 // n_dev + (1 if d < 11 % n_dev else 0) for d in range(n_dev)]
        assert pool.last_transport == ('rccl' if n_dev > 1 else 'upload (one device)')
        vn = vnd.VelvetNoise(sample_rate_hz=48000, seed=1)
        want = vn.decorrelate_batched(x)
        assert np.array_equal(want[3], vn.decorrelate(x[3]))
        for devices in ([0], 'all'):
            assert np.array_equal(vn.decorrelate_batched(x, devices=devices), want), devices
        mono = np.ascontiguousarray(x[:, :, :1])
        assert np.array_equal(vn.decorrelate_batched(mono, devices='all'), vn.decorrelate_batched(mono))
        with pytest.raises(ValueError):
            vnd.convolve_velvet_noise_batched(x, fir, devices=[n_dev])
        if n_dev > 1:                                               # the plain-upload transport gives the same tables
            up = multi.DevicePool(list(range(n_dev)), table_transport='upload')
            out = np.empty_like(x)
            from vndecorrelate_amd.taps import function_path_arrays
            up.map_streams(function_path_arrays(fir), x, out, 'convolve', vnd.MODE_EXACT)
            assert np.array_equal(out, vnd.convolve_velvet_noise_batched(x, fir)) and up.last_transport == 'upload'
            up.close()
    finally:
        multi.close_pools()


def test_single_process_rccl_communicator_carries_the_table():
    """The pool's table transport with more than one device - ncclCommInitAll in this process, one vnd_taps_broadcast_rccl
    per device - on as many devices as the box has (a one-device communicator on a one-GPU box: the root keeps its table,
    every RCCL call on the path still runs)."""
    import vndecorrelate_amd.decorrelation as vnd
    from vndecorrelate_amd import _native, multi
    from vndecorrelate_amd.taps import function_path_arrays
    from concurrent.futures import ThreadPoolExecutor
    n_dev = _native.device_count()
    arr = function_path_arrays(vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1))
    try:
        rccl = multi._Rccl()
    except RuntimeError:
        pytest.skip('librccl.so not found')
    comms = rccl.init_all(list(range(n_dev)))
    try:
        ctxs = [_native.context_for(d) for d in range(n_dev)]
        first = _native.TapTable.create(ctxs[0], arr.tap_offsets, arr.tap_index, arr.tap_weight)
        with ThreadPoolExecutor(n_dev) as ex:
            tables = list(ex.map(lambda r: _native.TapTable.broadcast_rccl(ctxs[r], first if r == 0 else None, 0, r, comms[r]), range(n_dev)))
        assert tables[0] is first and all(t.to_bytes() == arr.to_bytes() for t in tables)
    finally:
        rccl.destroy(comms)


def test_the_callers_current_device_is_left_alone():
    """A torch user calls the drop-in functions with a device list (or any entry point of a context on another device): the process's
    current HIP device afterwards is the one it was - vnd_ctx_create and the *_host entry points switch to their context's device
    inside a scope and restore it (round 5's advice: they used to leave the LAST device of the list current).  With one visible GPU
    the assertion is trivially true; with more it is taken on the last device while the work runs on the others."""
    import torch
    import vndecorrelate_amd.decorrelation as vnd
    from vndecorrelate_amd import _native, multi
    n_dev = _native.device_count()
    mine = n_dev - 1
    torch.cuda.set_device(mine)
    try:
        fir = vnd.generate_velvet_noise(duration_seconds=0.03, num_impulses=30, num_outs=2, sample_rate_hz=48000, seed=1)
        x = np.random.default_rng(5).uniform(-1, 1, (7, 20000, 2)).astype(np.float32)
        want = vnd.convolve_velvet_noise_batched(x, fir)
        for devices in ([0], 'all'):
            got = vnd.convolve_velvet_noise_batched(x, fir, devices=devices)
            assert np.array_equal(got, want)
            assert torch.cuda.current_device() == mine
        for d in range(n_dev):
            ctx = _native.context_for(d)
            assert ctx.device == d and torch.cuda.current_device() == mine
        y = vnd.VelvetNoise(sample_rate_hz=48000, seed=1).decorrelate_batched(x, devices='all')
        assert y.shape == x.shape and torch.cuda.current_device() == mine
    finally:
        multi.close_pools()
        torch.cuda.set_device(0)
