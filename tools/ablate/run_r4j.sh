#!/bin/bash
# cfg5 octets: the pad between the two quads' plane sets (VND_WIN_QUAD_PAD 0 / 16) - LDS bank conflicts and kernel time under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VND_TUNING=1
for pad in 0 16 0 16; do
  export VND_WIN_QUAD_PAD=$pad
  out=gpurun_out/r4j_pad$pad; mkdir -p $out
  timeout -k 5 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $out/sq -o p --output-format csv -- python3 tools/secondary_profile.py cfg5 40 > $out/sq.log 2>&1 || exit 1
  timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $out/trace -o p --output-format csv -- python3 tools/secondary_profile.py cfg5 40 > $out/trace.log 2>&1 || exit 1
  echo "== pad $pad"; python3 tools/summarize_profile.py $out 2>&1 | grep -i 'kernel trace\|CONFLICT\|IDX_ACTIVE\|INSTS_LDS\|conflicts'
  timeout -k 10 200 python tools/cfg5_try.py | cut -c1-140
done
