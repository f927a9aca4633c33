# pacing of a CU's two workgroups (VND_WIN_PACE): cfg3, cfg4 (N = 1 pass), the headline - off / on, interleaved
for p in 0 1 0 1; do
  echo "== VND_WIN_PACE=$p"
  VND_WIN_PACE=$p python tools/cfg5_try.py cfg3 2>&1 | grep "^cfg3" | cut -c1-90
  VND_TUNING=1 VND_WIN_PACE=$p python tools/shard_timeline.py 1024 100 2>&1 | grep "hipGraph"
  VND_TUNING=1 VND_WIN_PACE=$p python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['roofline']['kernel_ms'], d['roofline']['frac'], 'exact', d['config']['exact_mode']['kernel_ms'], d['config']['exact_mode']['frac_of_8TBs'], 'class', d['config']['exact_mode']['class_path_table']['kernel_ms'])"
done
