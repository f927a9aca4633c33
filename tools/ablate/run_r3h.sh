#!/bin/bash
# The transposition as planar chunks read back in 8-byte halves (lanes swizzled) against interleaved frame pairs read back in one
# 16-byte piece: LDS counters and the bench's kernel time.  bash tools/ablate/run_r3h.sh   (on the GPU box)
export TMPDIR=/tmp VND_TUNING=1 VND_SPEC_VERBOSE=1
for xp in 0 1; do
  out=gpurun_out/prof_xp$xp
  mkdir -p $out
  VND_WIN_XPOSE_PAIRS=$xp rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d $out -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-exact --no-secondary > $out.log 2>&1
  python3 - <<PY
import csv, collections, json
rows = list(csv.DictReader(open('$out/p_counter_collection.csv')))
acc = collections.defaultdict(list)
for r in rows:
    if r['Kernel_Name'] == 'vnd_spec_kernel':
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
line = [l for l in open('$out.log') if l.startswith('{"metric"')]
d = json.loads(line[-1]) if line else {}
print('frame pairs = $xp:', {k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, 'M per launch;', d.get('config', {}).get('launch', '')[0:130], 'parity', d.get('config', {}).get('parity_vs_oracle_of_peak'))
PY
  grep -i "spill\|unavailable" $out.log | head -3
done
for i in 1 2; do
  for xp in 0 1; do
    VND_WIN_XPOSE_PAIRS=$xp python3 bench.py --no-secondary --no-cpu > gpurun_out/bench_xp${xp}_$i.json 2>/dev/null
    python3 -c "
import json
d=json.loads(open('gpurun_out/bench_xp${xp}_$i.json').read().strip().splitlines()[-1])
print('pairs=$xp', d['roofline']['frac'], d['roofline']['kernel_ms'], d['config']['exact_mode']['frac_of_8TBs'], d['config']['exact_mode']['class_path_table']['frac_of_8TBs'], d['config']['launch'][:60])"
  done
done
