# store / load cache-policy bits of the per-table kernel on the headline config, interleaved
for rep in 1 2; do
for pol in "2 2" "3 2" "18 2" "19 2" "2 18" "2 3" "0 2"; do
  set -- $pol
  VND_SPEC_STORE_AUX=$1 VND_SPEC_LOAD_AUX=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary --no-exact 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('store aux $1 load aux $2', d['roofline']['frac'], d['roofline']['kernel_ms'])"
done
done
