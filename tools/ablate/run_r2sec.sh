# cfg3 / cfg5: kernel trace + PMC passes of the per-table kernel (what bounds the K >= 64 configs).
# Every pass under its own timeout: a counter set the hardware cannot collect makes rocprofv3 abort and then hang.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in cfg3 cfg5; do
  out=gpurun_out/prof_$cfg; mkdir -p $out
  run() { local name=$1; shift; timeout -k 5 150 rocprofv3 "$@" -d $out/$name -o p --output-format csv -- python3 tools/secondary_profile.py $cfg 20 > $out/$name.log 2>&1; echo "$cfg $name rc=$?"; }
  [ $cfg = cfg5 ] && run trace --kernel-trace --stats
  [ $cfg = cfg5 ] && run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
  [ $cfg = cfg5 ] && run sq2 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY
  run fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
  run write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  run tcc --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_READ_sum
  run ta --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
  run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
done
python3 tools/summarize_profile.py gpurun_out/prof_cfg3 > gpurun_out/prof_cfg3.txt 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_cfg5 > gpurun_out/prof_cfg5.txt 2>&1
echo done
