set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_spec.py -x -q -m gpu -k "exact_mode" > gpurun_out/pytest_r2_spec_exact.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_r2_spec_exact.log
timeout -k 10 300 python tools/exact_try.py > gpurun_out/exact_try.log 2>&1; echo exit $?; grep "ms/launch" gpurun_out/exact_try.log | cut -c1-200
