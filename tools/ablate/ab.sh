#!/bin/bash
# One parameterised A/B runner (replaces the per-experiment run_r?x.sh records; what those ran: RUNS.md).
#   bash tools/ablate/ab.sh [-n REPEATS] [-p "PMC COUNTERS ..."] [-o OUTSTEM] -- <probe command ...> -- "K=V K=V" "K=V" ...
# Runs the probe once per settings group (each group: space-separated KEY=VALUE pairs exported for that run, VND_TUNING=1 always), the
# groups interleaved REPEATS times (boxes drift: A B A B, never A A B B).  With -p the probe also runs once per group under
# `rocprofv3 --pmc <counters>` (its own pass, never combined with tracing) and tools/summarize_profile.py digests it.
# example:  bash tools/ablate/ab.sh -n 2 -- python tools/cfg5_try.py cfg3 -- "VND_WIN_PACE=1" "VND_WIN_PACE=0"
repeats=1; pmc=""; stem=gpurun_out/ab
while getopts "n:p:o:" opt; do case $opt in n) repeats=$OPTARG;; p) pmc=$OPTARG;; o) stem=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1)); [[ $1 == -- ]] && shift
cmd=(); while [[ $# -gt 0 && $1 != -- ]]; do cmd+=("$1"); shift; done; shift
groups=("$@"); [[ ${#groups[@]} -eq 0 ]] && groups=("")
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}" || exit 1
export VND_TUNING=1
for ((r = 0; r < repeats; ++r)); do
  for g in "${groups[@]}"; do
    echo "== [$r] ${g:-defaults}"
    # (the probe's own exit code decides, not grep's: a probe that prints nothing is not a failure, one that fails is)
    env $g timeout -k 10 300 "${cmd[@]}" 2>&1 | grep -v amdgpu.ids
    [[ ${PIPESTATUS[0]} -eq 0 ]] || { echo "probe failed (rc ${PIPESTATUS[0]})"; exit 1; }
  done
done
if [[ -n $pmc ]]; then
  k=0
  for g in "${groups[@]}"; do
    out=${stem}_pmc$k; mkdir -p "$out"; k=$((k + 1))
    # (the program itself after `--`: no env / bash -c hop under the profiler; the settings are exported into this shell instead)
    ( for kv in $g; do export "$kv"; done
      timeout -k 5 200 rocprofv3 --pmc $pmc -d "$out/sq" -o p --output-format csv -- "${cmd[@]}" > "$out/sq.log" 2>&1 ) || exit 1
    echo "== pmc ${g:-defaults}"; python3 tools/summarize_profile.py "$out" 2>&1 | grep -v '^ *[{}]' | head -40
  done
fi
