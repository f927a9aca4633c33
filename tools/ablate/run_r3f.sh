#!/bin/bash
# LDS bank conflicts of the headline kernel by the residue of the plane stride (16-byte slots between chunk planes, mod 8):
# which residues keep the 8-byte staging / read-back accesses conflict-free.  bash tools/ablate/run_r3f.sh   (on the GPU box)
export TMPDIR=/tmp VND_TUNING=1 VND_WIN_PLANE_MOD=8
for res in 2 1 3 4 5 6 7 0; do
  out=gpurun_out/prof_res$res
  mkdir -p $out
  VND_WIN_PLANE_RES=$res rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $out -o p --output-format csv -- python3 bench.py --steps 10 --warmup 3 --pool 512 --no-cpu --no-exact --no-secondary > $out.log 2>&1
  python3 - <<PY
import csv, collections, json
rows = list(csv.DictReader(open('$out/p_counter_collection.csv')))
acc = collections.defaultdict(list)
for r in rows:
    if r['Kernel_Name'] == 'vnd_spec_kernel':
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
line = [l for l in open('$out.log') if l.startswith('{"metric"')]
ms = json.loads(line[-1])['roofline']['kernel_ms'] if line else None
print('plane stride = $res mod 8 slots:', {k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, 'M per launch; kernel_ms under the profiler', ms)
PY
done
