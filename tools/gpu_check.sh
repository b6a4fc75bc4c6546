#!/bin/bash
# GPU-box smoke sequence: parity tests, default bench, 2-rank self-spawned bench, C5-shaped shard on one GPU.
# usage: tools/gpu_check.sh <tag>   (outputs under gpurun_out/<tag>/)
set -o pipefail
T=${1:-check}
O=gpurun_out/$T
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; tail -c 600 $O/bench_default.err; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gpus2.json 2> $O/bench_gpus2.err; rc=$?; tail -c 600 $O/bench_gpus2.err; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; rc=$?; tail -c 600 $O/bench_c5.err; [ $rc -eq 0 ] || exit $rc
echo ALL_OK
