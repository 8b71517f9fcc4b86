#!/bin/bash
# frame time against one environment knob of the library: env_sweep.sh <tag> <VAR> <values...>   (3-slice headline frame + the 1-slice frame)
tag=$1; var=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $out/b3_$v.json 2>> $out/err.txt || { echo "failed $v"; exit 1; }
  env $var=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --slices 1 > $out/b1_$v.json 2>> $out/err.txt || { echo "failed $v"; exit 1; }
  python3 -c "
import json; a=json.load(open('$out/b3_$v.json')); b=json.load(open('$out/b1_$v.json')); print('$var=$v: 3 slices', a['ms_per_step'], '/', a['other_semantics']['ms_per_step'], ' 1 slice', b['ms_per_step'], '/', b['other_semantics']['ms_per_step'], ' trace launch', b['roofline']['avg_launch_ms'])"
done
