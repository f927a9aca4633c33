# round 2: the artefacts committed under profiles/ (kernel trace + PMC passes of the bench, power table, RMS kernels)
mkdir -p gpurun_out; rm -f gpurun_out/pw4.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile.sh r02_spec > gpurun_out/profile_r02_spec.log 2>&1; echo "profile spec rc=$?"
VND_SPEC=0 bash tools/profile.sh r02_generic --pool 128 > gpurun_out/profile_r02_generic.log 2>&1; echo "profile generic rc=$?"
for v in spec generic spec; do
  unset VND_SPEC
  [ $v = generic ] && export VND_SPEC=0
  echo "== $v" >> gpurun_out/pw4.log
  timeout -k 10 120 python tools/power_probe.py 2 3 >> gpurun_out/pw4.log 2>&1 || exit 1
done
unset VND_SPEC
echo "== copy" >> gpurun_out/pw4.log; timeout -k 10 120 python tools/power_probe.py copy 3 >> gpurun_out/pw4.log 2>&1
python tools/summarize_power.py gpurun_out/pw4.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rms_rate3 -o p --output-format csv -- python3 tools/rms_rate.py > gpurun_out/prof_rms_rate3.log 2>&1; echo "rocprof rms rc=$?"; grep pool gpurun_out/prof_rms_rate3.log
