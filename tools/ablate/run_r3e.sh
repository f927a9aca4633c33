#!/bin/bash
# LDS bank conflicts of the headline kernel with the exact-fit ring (R = 301 entries) and with the ring rounded up to a multiple of
# 16 entries (R = 304): one --pmc pass each of bench.py's timed loop.  bash tools/ablate/run_r3e.sh   (on the GPU box)
export TMPDIR=/tmp VND_TUNING=1
for align in 0 1; do
  out=gpurun_out/prof_align$align
  mkdir -p $out
  VND_WIN_ALIGN_RING=$align rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $out -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-exact --no-secondary > $out.log 2>&1
  echo "align=$align rc=$?"
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('$out/p_counter_collection.csv')))
acc = collections.defaultdict(list)
for r in rows:
    if r['Kernel_Name'] == 'vnd_spec_kernel':
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('ring aligned = $align:', {k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, 'M per launch')
PY
done
