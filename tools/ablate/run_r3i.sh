export VND_TUNING=1 VND_SPEC_VERBOSE=1
for i in 1 2; do
for la in 4 3 2; do
  VND_SPEC_LA=$la python3 bench.py --no-secondary --no-cpu > gpurun_out/bench_la${la}_$i.json 2> gpurun_out/bench_la${la}_$i.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/bench_la${la}_$i.json').read().strip().splitlines()[-1])
print('LA=$la', d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['exact_mode']['frac_of_8TBs'], d['config']['exact_mode']['class_path_table']['frac_of_8TBs'], d['config']['launch'].split('threads=')[-1], '| exact', d['config']['exact_mode']['launch'].split('threads=')[-1])"
  grep -h "spills" gpurun_out/bench_la${la}_$i.err | sort | uniq -c | head -3
done
done
