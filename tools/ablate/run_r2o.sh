timeout -k 10 300 python tools/scan_breakdown.py 2>&1 | grep mode
for pool in 2048 4096; do timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary --no-exact --pool $pool 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pool', d['config']['pool_signals_per_gpu'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['timed_ms'])"; done
