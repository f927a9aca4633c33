set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_spec.py -x -q -m gpu -k "failed_runtime or out_of_scope or headline" > gpurun_out/pytest_r2_fb.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_r2_fb.log
export VND_BENCH_FORCE_DEVICE=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --pool 512 --no-exact > gpurun_out/bench_2rank_r2.json 2> gpurun_out/bench_2rank_r2.err; echo "2-rank bench exit $?"
python3 - <<'PY'
import json
for line in open('gpurun_out/bench_2rank_r2.json'):
    if line.startswith('{'):
        d=json.loads(line); print(d['n_gpus'], d['value'], d['roofline']['frac'], d['cfg4_strong'])
PY
tail -3 gpurun_out/bench_2rank_r2.err
