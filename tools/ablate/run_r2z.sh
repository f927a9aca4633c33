# round 2 final: refresh the artefacts under profiles/ with the final kernels, then the whole GPU tier and the bench
mkdir -p gpurun_out; rm -f gpurun_out/pw5.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile.sh r02_spec > gpurun_out/profile_r02_spec.log 2>&1; echo "profile spec rc=$?"
for v in spec copy; do
  echo "== $v" >> gpurun_out/pw5.log
  if [ $v = copy ]; then timeout -k 10 120 python tools/power_probe.py copy 3 >> gpurun_out/pw5.log 2>&1
  else timeout -k 10 120 python tools/power_probe.py 2 3 >> gpurun_out/pw5.log 2>&1 || exit 1; fi
done
python tools/summarize_power.py gpurun_out/pw5.log > gpurun_out/pw5.txt 2>&1; tail -8 gpurun_out/pw5.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rms_final -o p --output-format csv -- python3 tools/rms_rate.py > gpurun_out/prof_rms_final.log 2>&1; echo "rocprof rms rc=$?"; grep pool gpurun_out/prof_rms_final.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
