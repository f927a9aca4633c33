set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r2_full.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_r2_full.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r2_d.json 2> gpurun_out/bench_r2_d.err; echo "bench exit $?"; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2_d.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'][:240])
print(d['cfg4_strong']['ms_per_pass_max_over_ranks'], {k:(v.get('kernel_ms'),v.get('achieved_GBs')) for k,v in d['secondary'].items()})
print(d['config']['exact_mode'])
PY
for pool in 128 256; do timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu --no-secondary --no-exact --pool $pool 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool', d['config']['pool_signals_per_gpu'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'][30:220])"; done
