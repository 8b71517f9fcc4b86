#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-stripe_n1}; mkdir -p $out
for rep in 1 2; do for sr in 8 4 2 1; do
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --stripe-rows $sr > $out/b_$sr.json 2>> $out/err.txt || { echo failed; exit 1; }
  python3 -c "
import json; a=json.load(open('$out/b_$sr.json')); print('stripe rows $sr: N=1 frame', a['ms_per_step'], '/', a['other_semantics']['ms_per_step'])"
done; done
