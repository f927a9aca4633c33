#!/bin/bash
# Profile the bench workload with rocprofv3: one kernel-trace pass (+stats) and
# separate PMC passes (never combined with API tracing).  Run on the GPU box:
#   bash tools/profile.sh <tag> [bench args...]
# Raw output goes to gpurun_out/prof_<tag>/, the digest to gpurun_out/prof_<tag>.json/.txt
set -u
tag=${1:-run}; shift || true
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=(--steps 20 --warmup 5 --no-cpu --no-exact --no-secondary --no-strong --no-power "$@")
run() {  # name, rocprof options...
  local name=$1; shift
  rocprofv3 "$@" -d "$out/$name" -o p --output-format csv -- python3 bench.py "${args[@]}" > "$out/$name.log" 2>&1
  echo "$name rc=$?"
}
# (the trace pass times 100 launches, so that the csv's plain average - which includes the cold warm-up launches -
#  is the steady-state figure to within ~1 %)
args_pmc=("${args[@]}")
args=(--steps 100 --warmup 5 --no-cpu --no-exact --no-secondary --no-strong --no-power "$@")
run trace --kernel-trace --stats
args=("${args_pmc[@]}")
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_WAVES_EQ_64
run fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcc --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_READ_sum
python3 tools/summarize_profile.py "$out" > "gpurun_out/prof_$tag.txt" 2>&1
cat "gpurun_out/prof_$tag.txt"
