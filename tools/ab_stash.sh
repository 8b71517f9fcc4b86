#!/bin/bash
# A/B of the speculative leaf postponement (MI355RT_STASH) on the GPU box: parity subset with the variant on,
# then frame / trace times with lane-utilisation counters for several leaf thresholds.
out=gpurun_out/${1:-ab_stash}; mkdir -p $out
MI355RT_STASH=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closest_hit or film_bit_exact or per_node or shadow_predicate or full_size_matches or large_random or other_recursion or axis_parallel" > $out/pytest_stash.log 2>&1; echo "parity with STASH: exit $?"; tail -2 $out/pytest_stash.log
for cfg in "0 16" "1 16" "1 24" "1 32" "1 40" "1 48"; do
  set -- $cfg
  echo "== STASH=$1 LEAF_THRESHOLD=$2"
  MI355RT_STASH=$1 MI355RT_LEAF_THRESHOLD=$2 MI355RT_SLICES=1 MI355RT_DEBUG_UTIL=1 ITERS=2 timeout -k 10 120 python tools/gpu_explore.py thai2 16 6 2>&1 | grep -E "spp=|lane util" | tail -2
  MI355RT_STASH=$1 MI355RT_LEAF_THRESHOLD=$2 ITERS=4 timeout -k 10 120 python tools/gpu_explore.py thai2,ico2,4boxes 64 0 2>&1 | grep -E "spp=" | awk 'NR%4==0'
done 2>&1 | tee $out/ab.txt
