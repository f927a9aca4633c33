set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r2_full.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_r2_full.log
timeout -k 10 300 python tools/scan_rate.py > gpurun_out/scan_r2.log 2>&1; echo "scan exit $?"; cat gpurun_out/scan_r2.log | grep -v amdgpu
timeout -k 10 300 python tools/sweep.py --config cfg5 --modes 2 --rs 4 --rounds 3 > gpurun_out/sweep_cfg5_r2.log 2>&1; grep fast gpurun_out/sweep_cfg5_r2.log
timeout -k 10 300 python tools/sweep.py --config cfg5 --modes 2 --rs 4 --rounds 3 --iters 200 > gpurun_out/sweep_cfg5_r2b.log 2>&1; grep fast gpurun_out/sweep_cfg5_r2b.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_r2_b.json 2> gpurun_out/bench_r2_b.err; echo "bench exit $?"; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2_b.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'][:120])
print(d['cfg4_strong']['ms_per_pass_max_over_ranks'], {k:(v.get('kernel_ms'),v.get('achieved_GBs')) for k,v in d['secondary'].items()})
print(d['end_to_end'])
PY
