set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_spec.py -x -q -m gpu -k "geometries or headline or out_of_scope or real_size or linearity or g48k_k30" > gpurun_out/pytest_r2_spec.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_r2_spec.log
for pool in 2048 1024 3072 128; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary --no-exact --pool $pool 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool', d['config']['pool_signals_per_gpu'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['launch'][100:220])"; done
