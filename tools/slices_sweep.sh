#!/bin/bash
# headline frame against the number of concurrent frame slices
out=gpurun_out/${1:-slices}; mkdir -p $out; cd $GRAFT_REPO_ROOT
for sl in 1 2 3 4 5 6; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --slices $sl > $out/b_$sl.json 2>> $out/err.txt || { echo "failed $sl"; exit 1; }
  python3 -c "
import json; a=json.load(open('$out/b_$sl.json')); print('slices $sl: frame', a['ms_per_step'], '/', a['other_semantics']['ms_per_step'])"
done
