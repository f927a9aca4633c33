#!/bin/bash
# cfg5 octets: the ring cut to a multiple of 16 entries with a TAIL behind the mirror (VND_WIN_TAIL 0 / 1) - parity, time, LDS bank conflicts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VND_TUNING=1
for tail in 0 1 0 1; do
  echo "== VND_WIN_TAIL=$tail"; timeout -k 10 200 python tools/cfg5_try.py VND_WIN_TAIL=$tail | cut -c1-150 || exit 1
done
for tail in 0 1; do
  export VND_WIN_TAIL=$tail
  out=gpurun_out/r4l_tail$tail; mkdir -p $out
  timeout -k 5 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $out/sq -o p --output-format csv -- python3 tools/secondary_profile.py cfg5 40 > $out/sq.log 2>&1 || exit 1
done
unset VND_WIN_TAIL
timeout -k 10 600 python -m pytest tests/test_gpu_win.py tests/test_gpu_fuzz.py -x -q -k "quads or octets or bench_shapes or random" | tail -3
