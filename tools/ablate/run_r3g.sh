#!/bin/bash
# LDS bank conflicts of the headline kernel without / with the lanes' pair-index swizzle (planes an odd number of slots apart /
# 2 mod 4 apart + bits 3 and 4 of the pair index exchanged).  bash tools/ablate/run_r3g.sh   (on the GPU box)
export TMPDIR=/tmp VND_TUNING=1
for swz in 0 1; do
  out=gpurun_out/prof_swz$swz
  mkdir -p $out
  VND_WIN_LANE_SWIZZLE=$swz rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $out -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-exact --no-secondary > $out.log 2>&1
  python3 - <<PY
import csv, collections, json
rows = list(csv.DictReader(open('$out/p_counter_collection.csv')))
acc = collections.defaultdict(list)
for r in rows:
    if r['Kernel_Name'] == 'vnd_spec_kernel':
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
line = [l for l in open('$out.log') if l.startswith('{"metric"')]
d = json.loads(line[-1]) if line else {}
print('lane swizzle = $swz:', {k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, 'M per launch;', d.get('config', {}).get('launch', '')[40:130], 'parity', d.get('config', {}).get('parity_vs_oracle_of_peak'))
PY
done
