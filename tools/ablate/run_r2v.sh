for aux in 0 2 0 2 1 16; do
  VND_SPEC_LOAD_AUX=$aux timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary --no-exact 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('load aux $aux', d['roofline']['frac'], d['roofline']['kernel_ms'])"
done
