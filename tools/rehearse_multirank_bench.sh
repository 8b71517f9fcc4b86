#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: 2 ranks share cuda:0, torch.distributed over gloo
# (MI355RT_BENCH_SHARE_GPU=1); bench.py launches its two ranks itself.  RCCL itself cannot be rehearsed this way (it refuses two ranks on one device).
export MI355RT_BENCH_SHARE_GPU=1
run() { timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline "${@:2}" 2>&1 | grep -E '^\{|Error|error|Traceback' | cut -c1-700; }
echo "== weak, torch gather"; run 29511 --native-gather off
echo "== strong, torch gather"; run 29512 --native-gather off --scaling strong
echo "== c5 strong (4 spp), torch gather"; run 29513 --native-gather off --scaling strong --config c5 --spp 4
echo "== weak, native gather attempted (expected: falls back, RCCL refuses two ranks on one GPU)"; run 29514 --native-gather auto
