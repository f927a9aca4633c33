# does the run length move the headline? (power/thermal averaging) - interleaved on one box
for k in 20 40 20 40 80 20; do
  timeout -k 10 300 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu --no-secondary --no-exact 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $k', d['roofline']['frac'], d['roofline']['kernel_ms'], d['timed_ms'], d['warmup_actual'])"
done
