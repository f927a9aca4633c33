"""CPU tier: the shape of bench.py's stdout line and its self-launch of N ranks (no GPU, no torch import in the parent)."""
import io
import json
import pathlib
import sys

REPO = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def _detail():
    """A committed full record of a real run (the newest profiles/r0N_bench_detail.json; round 4's line had the same legs)."""
    found = sorted(REPO.glob('profiles/r0*_bench_detail.json'))
    d = json.loads((found[-1] if found else REPO / 'profiles' / 'r04_bench_line.json').read_text())
    d['config'].setdefault('world_size', 1)
    d['config'].setdefault('backend', None)
    d['config'].setdefault('ranks_seen_by_all_reduce', 1)
    d['roofline'].setdefault('power', None)
    d['cpu_baseline'].setdefault('sample_short', d['cpu_baseline']['sample'][:100])
    return d


def _strings(o):
    if isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _strings(v)
    elif isinstance(o, str):
        yield o


def test_the_line_is_short_and_carries_every_claim():
    import bench
    d = _detail()
    line = bench.compact(d)
    text = json.dumps(line, separators=(',', ':'))
    assert len(text) < 4096                                            # the driver keeps a tail of the output
    assert all(len(s) <= 120 for s in _strings(line))                  # and cuts strings at 120 characters
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert set(line['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'power_W', 'sclk_MHz'}
    assert set(line['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    config = line['config']
    assert 'model' not in config and config['workload'].startswith('cfg2')
    # one number per row of README's table
    for key in ('exact_frac', 'class_exact_frac', 'cfg3_frac', 'cfg3_fp32_frac', 'cfg3k1_frac', 'cfg5_frac', 'cfg5_exact_frac', 'cfg4_frac',
                'f1_256_exact_frac', 'f1_256_fast_frac', 'f1_128_exact_frac', 'm2s_frac', 'm2s_exact_frac', 'scan_ms', 'chain_ms',
                'e2e_cfg2_ms', 'e2e_cfg4_ms', 'proj_N1_us', 'proj_N8_us', 'proj_speedup', 'worst_parity_over_pools', 'single_us'):
        assert isinstance(config[key], (int, float)), key
    assert config['exact_frac'] == d['exact_mode']['frac_of_8TBs']
    assert config['cfg3_frac'] == d['secondary']['cfg3']['frac_of_8TBs']
    assert config['cfg4_strong']['ranks'] == d['cfg4_strong']['ranks']
    assert config['worst_parity_over_pools'] <= 1e-6


def test_a_failed_leg_costs_its_keys_only():
    import bench
    d = _detail()
    d['secondary']['cfg5'] = {'error': 'RuntimeError("x")'}
    del d['next_rows']
    d['exact_mode'] = None
    config = bench.compact(d)['config']
    assert 'cfg5_frac' not in config and 'cfg5_error' in config and config['exact_frac'] is None and config['cfg3_frac'] > 0


def test_plain_gpus_n_starts_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus 4` without WORLD_SIZE: a CHILD `python -m torch.distributed.run` with the same arguments; rank 0's line
    is relayed to stdout, other output to stderr, the child's code is the exit code; the parent does not import torch."""
    import subprocess
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, **kw):
            seen['cmd'], seen['kw'] = cmd, kw
            self.stdout = io.StringIO('W0101 torchrun chatter\n{"metric": "m", "value": 1}\n')

        def wait(self):
            return 7

    monkeypatch.setattr(subprocess, 'Popen', FakeProc)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '2'])
    for name in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(name, raising=False)
    had_torch = 'torch' in sys.modules
    try:
        bench.main()
        raise AssertionError('main() returned')
    except SystemExit as e:
        assert e.code == 7
    assert had_torch or 'torch' not in sys.modules
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--master-addr' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '4', '--steps', '2'] and cmd[-5].endswith('bench.py')
    assert seen['kw']['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    out = capsys.readouterr()
    assert out.out == '{"metric": "m", "value": 1}\n' and 'chatter' in out.err
