for aux in 2 0 2 0; do
  VND_SPEC_LOAD_AUX=$aux timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-exact > gpurun_out/w_$aux.log 2> gpurun_out/w_$aux.err
  tail -1 gpurun_out/w_$aux.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('load aux $aux', d['roofline']['frac'], d['roofline']['kernel_ms']); print({k:(v['kernel_ms'],v['achieved_GBs']) for k,v in d['secondary'].items()}); print(d['cfg4_strong']['ms_per_pass_max_over_ranks'])" || tail -5 gpurun_out/w_$aux.err
done
