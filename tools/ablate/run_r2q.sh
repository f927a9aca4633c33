set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r2_full.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_r2_full.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r2_e.json 2> gpurun_out/bench_r2_e.err; echo "bench exit $?"; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2_e.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['config']['launch'][:60])
print(d['cfg4_strong']['ms_per_pass_max_over_ranks'], {k:(v.get('kernel_ms'),v.get('achieved_GBs'),v.get('launch','')[:24]) for k,v in d['secondary'].items()})
print(d['config']['exact_mode']['kernel_ms'], d['end_to_end']['cfg4_batch'])
PY
