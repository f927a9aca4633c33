#!/usr/bin/env python3
"""Turn a tools/profile.sh digest (gpurun_out/prof_<tag>.txt + the raw kernel_stats.csv) into the
committed summaries under profiles/:  publish_profile.py <tag> <kernel substring> <out stem>"""
import glob, json, pathlib, sys

REPO = pathlib.Path(__file__).resolve().parents[1]
tag, needle, stem = sys.argv[1], sys.argv[2], sys.argv[3]
digest = json.loads((REPO / 'gpurun_out' / f'prof_{tag}.txt').read_text())
kernel = next(k for k in digest['kernels'] if needle in k)
trace = digest['kernels'][kernel]
counters = {c: round(v['mean'], 1) for c, v in digest['counters'].get(kernel, {}).items()}
out = {
    'command': 'rocprofv3 {--kernel-trace --stats | --pmc <one group per pass>} -- python3 bench.py --steps 20 '
               '--warmup 5 --no-cpu --no-exact --no-secondary [--mode exact]  (tools/profile.sh)',
    'kernel': kernel,
    'kernel_trace_us': trace,
    'counters_mean_per_dispatch': dict(sorted(counters.items())),
    'notes': ['SQ_* cycle counters are in quad-cycles summed over all waves/SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs',
              'FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream '
              '(MI355X_MICROARCH.md, HBM section): doubled below'],
}
c = counters
# the bench line of the traced pass tells the pool size and the launch geometry
bench_line = {}
for line in (REPO / 'gpurun_out' / f'prof_{tag}' / 'trace.log').read_text().splitlines():
    if line.startswith('{"metric"'):          # stderr carries the full record before stdout's short line: the one with the launch text
        cand = json.loads(line)
        if 'launch' in cand.get('config', {}) or not bench_line:
            bench_line = cand
pool = bench_line.get('config', {}).get('pool_signals_per_gpu', 128)
launch = bench_line.get('config', {}).get('launch', '')
out['bench_line_of_traced_pass'] = {k: bench_line.get(k) for k in ('value', 'ms_per_step', 'steps', 'warmup_actual')}
out['bench_line_of_traced_pass']['kernel_ms'] = bench_line.get('roofline', {}).get('kernel_ms')
out['launch'] = launch
derived = {'algorithmic_bytes_per_launch': 8 * pool * 480000 * 2, 'pool_signals': pool}
if 'FETCH_SIZE' in c:
    derived['hbm_read_bytes_per_launch'] = c['FETCH_SIZE'] * 1024 * 2
if 'WRITE_SIZE' in c:
    derived['hbm_write_bytes_per_launch'] = c['WRITE_SIZE'] * 1024
if 'GRBM_GUI_ACTIVE' in c:
    derived['effective_clock_GHz'] = round(c['GRBM_GUI_ACTIVE'] / 8 / (trace['mean_us'] * 1e-6) / 1e9, 3)
if 'SQ_ACTIVE_INST_VALU' in c and 'SQ_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
    # quad-cycles summed over 1024 SIMDs vs the chip's busy cycles
    cycles = c['GRBM_GUI_ACTIVE'] / 8
    derived['valu_busy_frac'] = round(c['SQ_ACTIVE_INST_VALU'] * 4 / (cycles * 1024), 3)
    if 'SQ_LDS_IDX_ACTIVE' in c:
        derived['lds_busy_frac'] = round(c['SQ_LDS_IDX_ACTIVE'] / (cycles * 256), 3)
if 'SQ_WAVES' in c:
    for name in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM'):
        if name in c:
            derived[name.lower().replace('sq_', '') + '_per_wave'] = round(c[name] / c['SQ_WAVES'], 1)
out['derived'] = derived
(REPO / 'profiles' / f'{stem}_pmc.json').write_text(json.dumps(out, indent=1) + '\n')
if 'hbm_read_bytes_per_launch' in derived and 'hbm_write_bytes_per_launch' in derived and '--no-traffic' not in sys.argv:
    total = derived['hbm_read_bytes_per_launch'] + derived['hbm_write_bytes_per_launch']
    (REPO / 'profiles' / 'hbm_traffic.json').write_text(json.dumps({
        # kernel + tile geometry, as vnd_describe_launch prints it (without the read depth: under rocprofv3 hipRTC's build of the
        # same source keeps 4 reads ahead without spilling, outside it the library settles for 3 - the bytes moved are the same)
        'kernel': launch.split(' prefetch=')[0].split(' mode=')[0].split(' reads_ahead=')[0],
        'pool': pool, 'bytes_per_launch': int(total), 'bytes_per_stream': total / pool,
        'read_bytes': int(derived['hbm_read_bytes_per_launch']), 'write_bytes': int(derived['hbm_write_bytes_per_launch']),
        'algorithmic_bytes_per_stream': 8 * 480000 * 2, 'launch': launch,
        'commit': __import__('subprocess').run(['git', 'rev-parse', '--short', 'HEAD'], cwd=str(REPO), capture_output=True, text=True).stdout.strip() or 'unknown',
        'source': f'profiles/{stem}_pmc.json: FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate --pmc passes'}, indent=1) + '\n')
stats = glob.glob(str(REPO / 'gpurun_out' / f'prof_{tag}' / 'trace' / '**' / '*kernel_stats.csv'), recursive=True)
if stats:
    lines = pathlib.Path(stats[0]).read_text().splitlines()
    keep = [lines[0]] + [l for l in lines[1:] if 'vnd' in l or 'copyBuffer' in l or 'stream_copy_kernel' in l]      # vnd:: kernels, the hipRTC-built vnd_spec_kernel, the copies beside them
    (REPO / 'profiles' / f'{stem}_kernel_stats.csv').write_text('\n'.join(keep) + '\n')
print(json.dumps({'kernel': kernel, 'trace': trace, 'derived': derived}, indent=1))
