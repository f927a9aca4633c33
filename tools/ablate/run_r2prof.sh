cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile.sh r02_spec > gpurun_out/profile_r02_spec.log 2>&1; echo "profile spec rc=$?"
