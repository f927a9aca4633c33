#!/bin/bash
# A round's committed profiles in one call (replaces the per-round tools/ablate/run_r?prof.sh):
#   bash tools/profile_round.sh <tag>            e.g. r05  ->  gpurun_out/prof_<tag>*; copy the digests into profiles/<tag>_*
# headline: kernel trace + one --pmc pass per counter group of bench.py (tools/profile.sh), then the same groups for cfg3 and cfg5
# (tools/secondary_profile.py), then the kernel traces of the f1 pool stage and of the N = 8 shard.  One counter group per
# rocprofv3 pass, never combined with API tracing.  What to skip: SKIP="f1 shard" etc.
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}      # (the GPU box exports GRAFT_REPO_ROOT; anywhere else: this script's repository)
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
skip=" ${SKIP:-} "
if [[ $skip != *" headline "* ]]; then bash tools/profile.sh "$tag" > "gpurun_out/prof_$tag.log" 2>&1; echo "headline rc=$?"; fi
for cfg in cfg3 cfg5; do
  [[ $skip == *" $cfg "* ]] && continue
  out=gpurun_out/prof_${tag}_$cfg; mkdir -p "$out"
  run() { local name=$1; shift; timeout -k 5 150 rocprofv3 "$@" -d "$out/$name" -o p --output-format csv -- python3 tools/secondary_profile.py $cfg 40 > "$out/$name.log" 2>&1; echo "$cfg $name rc=$?"; }
  run trace --kernel-trace --stats
  run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
  run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
  run sq3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INST_LEVEL_LDS
  run fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
  run write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  python3 tools/summarize_profile.py "$out" > "gpurun_out/prof_${tag}_$cfg.txt" 2>&1
done
if [[ $skip != *" f1 "* ]]; then
  for p in 256 128; do
    timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "gpurun_out/prof_${tag}_f1_$p" -o p --output-format csv -- python3 tools/f1_pool_rate.py $p > "gpurun_out/prof_${tag}_f1_$p.log" 2>&1; echo "f1 $p rc=$?"
    python3 tools/trace_gaps.py "gpurun_out/prof_${tag}_f1_$p" > "gpurun_out/prof_${tag}_f1_$p.txt" 2>&1
  done
fi
if [[ $skip != *" shard "* ]]; then
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "gpurun_out/prof_${tag}_shard" -o p --output-format csv -- python3 tools/shard_timeline.py 128 300 > "gpurun_out/prof_${tag}_shard.log" 2>&1; echo "shard rc=$?"
  python3 tools/trace_gaps.py "gpurun_out/prof_${tag}_shard" > "gpurun_out/prof_${tag}_shard.txt" 2>&1
fi
