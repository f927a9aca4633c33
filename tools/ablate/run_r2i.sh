set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_spec.py -x -q -m gpu -k "geometries or headline or out_of_scope or real_size or linearity" > gpurun_out/pytest_r2_spec.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_r2_spec.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_r2_c.json 2> gpurun_out/bench_r2_c.err; echo "bench exit $?"; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2_c.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'][:200])
print(d['cfg4_strong']['ms_per_pass_max_over_ranks'], {k:(v.get('kernel_ms'),v.get('achieved_GBs')) for k,v in d['secondary'].items()})
PY
for pool in 128 512 1024; do timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu --no-secondary --no-exact --pool $pool 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool', d['config']['pool_signals_per_gpu'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'][100:200])"; done
