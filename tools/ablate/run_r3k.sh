#!/bin/bash
# LDS counters of the pair-read per-table kernel on cfg5 (8 channels) and of the window kernel on cfg3: where bank conflicts are left.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg5 cfg3; do
  out=gpurun_out/prof_r3k_$cfg; mkdir -p $out
  timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES -d $out -o p --output-format csv -- python3 tools/secondary_profile.py $cfg 30 > $out.log 2>&1
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('$out/p_counter_collection.csv')))
acc = collections.defaultdict(list)
for r in rows:
    if r['Kernel_Name'] == 'vnd_spec_kernel':
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$cfg', {k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, 'M per launch', open('$out.log').read().splitlines()[0][:120])
PY
done
