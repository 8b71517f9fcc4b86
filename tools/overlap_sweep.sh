#!/bin/bash
# multi-slice frame time: lockstep slices (MI355RT_NO_PIPELINE) against the pipelined schedule, by the occupancy the trace
# kernel is launched with (MI355RT_PIPE_BLOCKS blocks per CU, 0 = all it can get) and the slice count
out=gpurun_out/${1:-overlap}; mkdir -p $out
cd $GRAFT_REPO_ROOT
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --slices $sl > $out/b_${sl}_${label}.json 2>> $out/err.txt || { echo "failed $sl $label"; exit 1; }
  python3 -c "
import json; d=json.load(open('$out/b_${sl}_${label}.json')); print('slices $sl $label: frame', d['ms_per_step'], 'ms | tch', d['other_semantics']['ms_per_step'])"
}
for sl in ${SLICES:-3 2 4}; do
  run lockstep MI355RT_NO_PIPELINE=1
  for b in ${BLOCKS:-0 6 5 4 3}; do run pipe_$b MI355RT_PIPE_BLOCKS=$b; done
done
