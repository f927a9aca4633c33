cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "sums or rms or decorrelate or stage" 2>&1 | tail -2 || exit 1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rms5 -o p --output-format csv -- python3 tools/rms_rate.py > gpurun_out/prof_rms5.log 2>&1
grep "pool    1" gpurun_out/prof_rms5.log
