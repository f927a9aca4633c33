#!/bin/bash
# round 3 (late): PMC passes of the window form on channel OCTETS / QUADS (vw_span_q) on cfg5, one counter group per pass.
#   run_r3q.sh octet            (the default launch)
#   VND_WIN_OCTET=0 run_r3q.sh quad        VND_WIN_QUAD=0 run_r3q.sh pairs   (the pair-read kernel beside them)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-octet}
out=gpurun_out/prof_r3_c8_$tag; mkdir -p $out
run() { local name=$1; shift; timeout -k 5 150 rocprofv3 "$@" -d $out/$name -o p --output-format csv -- python3 tools/secondary_profile.py cfg5 40 > $out/$name.log 2>&1; echo "cfg5 $tag $name rc=$?"; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INST_LEVEL_LDS
run fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcc --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_READ_sum
python3 tools/summarize_profile.py $out > gpurun_out/prof_r3_c8_$tag.txt 2>&1
