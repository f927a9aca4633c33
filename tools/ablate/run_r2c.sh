set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r2_full.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_r2_full.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r2_a.json 2> gpurun_out/bench_r2_a.err; echo "bench exit $?"; cat gpurun_out/bench_r2_a.json; tail -3 gpurun_out/bench_r2_a.err
