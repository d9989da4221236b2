#!/bin/bash
# two ranks of bench.py on ONE GPU over gloo: exercises the N > 1 code path end to end (shards, all-gathers, max-over-ranks timing)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --dist-backend gloo --single-device > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo rc=$?; tail -5 gpurun_out/bench_2rank.err | cut -c1-300; cut -c1-600 gpurun_out/bench_2rank.json
