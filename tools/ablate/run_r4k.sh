#!/bin/bash
# instruction cache: requests / hits / misses and the fetch level of the per-table window kernels (cfg5's octet kernel is ~120 KB of
# straight-line code, eight different functions for the eight waves of a workgroup; cfg3's ~48 KB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg5 cfg3; do
  out=gpurun_out/r4k_$cfg; mkdir -p $out
  timeout -k 5 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $out/ic -o p --output-format csv -- python3 tools/secondary_profile.py $cfg 40 > $out/ic.log 2>&1 || exit 1
  timeout -k 5 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d $out/w -o p --output-format csv -- python3 tools/secondary_profile.py $cfg 40 > $out/w.log 2>&1 || exit 1
  echo "== $cfg"; python3 tools/summarize_profile.py $out 2>&1 | grep -v '^ *[{}]' | head -40
done
