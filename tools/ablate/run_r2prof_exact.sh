cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile.sh r02_exact --mode exact > gpurun_out/profile_r02_exact.log 2>&1; echo "profile exact rc=$?"
timeout -k 10 200 python tools/rms_batch_rate.py 2>&1 | tail -6
