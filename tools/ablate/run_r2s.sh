set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_spec.py -x -q -m gpu -k "lopsided or class_path" > gpurun_out/pytest_r2_lop.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_r2_lop.log
timeout -k 10 500 python tools/big_stream_check.py > gpurun_out/big_r2.log 2>&1; echo "big exit $?"; grep -v amdgpu gpurun_out/big_r2.log
timeout -k 10 200 python tools/scan_rate.py 2>&1 | grep "device scan"
