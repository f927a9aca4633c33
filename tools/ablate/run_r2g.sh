set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r2_full.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_r2_full.log
echo "--- round-1 tree, cfg5 and cfg3:"
(cd tmp_r1 && timeout -k 10 300 python tools/sweep.py --config cfg5 --modes 2 --rs 4 --rounds 3 2>&1 | grep fast; timeout -k 10 300 python tools/sweep.py --config cfg5 --modes 2 --rs 4 --rounds 3 --iters 200 2>&1 | grep fast)
echo "--- this tree:"
timeout -k 10 300 python tools/sweep.py --config cfg5 --modes 2 --rs 4 --rounds 3 2>&1 | grep fast
