#!/bin/bash
# headline frame with the BVH built on the device against the host SAH build, by the leaf size of the device tree
out=gpurun_out/${1:-lbvh}; mkdir -p $out; cd $GRAFT_REPO_ROOT
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc $EXTRA > $out/b_$label.json 2>> $out/err.txt || { echo "failed $label"; exit 1; }
  python3 -c "
import json; a=json.load(open('$out/b_$label.json')); r=a['roofline']; ac=a['accel']; print('$label: frame', a['ms_per_step'], '/', a['other_semantics']['ms_per_step'], ' nodes/ray', r['nodes_per_ray'], 'tris/ray', r['tris_per_ray'], '|', ac['builder'], 'nodes', ac['nodes'], 'depth', ac['max_depth'], 'build wall', ac['bvh_build_ms'], 'device', ac['device_build_ms'])"
}
EXTRA=""; run host_sah MI355RT_X=0
EXTRA="--device-lbvh"; for l in 1 2 3 4; do run lbvh_leaf$l MI355RT_MAX_LEAF=$l; done
