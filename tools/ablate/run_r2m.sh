set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_spec.py -x -q -m gpu > gpurun_out/pytest_r2_spec.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_r2_spec.log
timeout -k 10 300 python tools/spec_try.py 0.8 cfg2 > gpurun_out/spec_try6_cfg2.log 2>&1; echo exit $?; grep -E "ms/launch|vs" gpurun_out/spec_try6_cfg2.log | cut -c1-170 | sed -e "s/conv_spec (hipRTC, per table)//"
timeout -k 10 400 python tools/spec_try.py 0.8 cfg3 > gpurun_out/spec_try6_cfg3.log 2>&1; echo exit $?; grep -E "ms/launch|vs" gpurun_out/spec_try6_cfg3.log | cut -c1-170 | sed -e "s/conv_spec (hipRTC, per table)//"
