cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "sums or rms or decorrelate or stage or exact" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python tools/rms_ties_rate.py 2>&1 | tail -8
