#!/bin/bash
# nodes / triangles per ray and the frame time against one environment knob: nodes_probe.sh <tag> <VAR> <values...>
tag=$1; var=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $out/b_$v.json 2>> $out/err.txt || { echo "failed $v"; exit 1; }
  python3 -c "
import json; a=json.load(open('$out/b_$v.json')); r=a['roofline']; print('$var=$v: frame', a['ms_per_step'], '/', a['other_semantics']['ms_per_step'], ' nodes/ray', r['nodes_per_ray'], 'tris/ray', r['tris_per_ray'], 'trace launch', r['avg_launch_ms'])"
done
