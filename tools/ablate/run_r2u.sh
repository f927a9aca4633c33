set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r2_full.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_r2_full.log
timeout -k 10 500 python tools/big_dec_probe.py 2>&1 | grep -v amdgpu | head -3
