#!/bin/bash
# round 3: instruction-fetch and LDS-queue counters of the window form on cfg3 and cfg2 (is the 48 KB straight-line tap code an I-cache problem?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg3 cfg2; do
  out=gpurun_out/prof_r3d_$cfg; mkdir -p $out
  n=40; [ $cfg = cfg2 ] && n=100
  timeout -k 5 150 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $out/if -o p --output-format csv -- python3 tools/secondary_profile.py $cfg $n > $out/if.log 2>&1; echo "$cfg if rc=$?"
  timeout -k 5 150 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $out/lv -o p --output-format csv -- python3 tools/secondary_profile.py $cfg $n > $out/lv.log 2>&1; echo "$cfg lv rc=$?"
  python3 tools/summarize_profile.py $out > gpurun_out/prof_r3d_$cfg.txt 2>&1
done
