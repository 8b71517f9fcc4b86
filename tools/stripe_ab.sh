#!/bin/bash
# stripe height 8 against 4: the N = 1 frame (blocks dealt to the slices) and rank 0's share of N-GPU frames
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-stripe}; mkdir -p $out
for sr in 8 4; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --stripe-rows $sr > $out/b_$sr.json 2>> $out/err.txt || { echo failed; exit 1; }
  python3 -c "
import json; a=json.load(open('$out/b_$sr.json')); print('stripe rows $sr: N=1 frame', a['ms_per_step'], '/', a['other_semantics']['ms_per_step'])"
  PROBE_STRIPE_ROWS=$sr timeout -k 10 400 python tools/weak_scaling_probe.py 2>&1 | grep -E "weak|strong 1080p" 
done
