#!/bin/bash
# kernel stats of two 1-slice frames with experiment builds of the kernels: variant_stats.sh <tag> <variant> [<variant> ...]
# (variants: raytracer-rs_amd/libmi355rt_<variant>.so from `make variant`; "base" = the shipped library)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  out=gpurun_out/$tag/$v; mkdir -p $out
  if [ "$v" = base ]; then unset MI355RT_LIB; else export MI355RT_LIB=$GRAFT_REPO_ROOT/raytracer-rs_amd/libmi355rt_$v.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --slices 1 > $out/bench_1slice.json 2> $out/err.txt || { echo "$v failed"; exit 1; }
  echo "== $v"
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$out/stats/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.4:
        print("%-58s calls %4s avg %8.1f us  %5.1f%%" % (r["Name"].replace("mi355rt::", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $out/bench.json 2>> $out/err.txt
  python3 -c "
import json; d=json.load(open('$out/bench.json')); print('frame (3 slices)', d['ms_per_step'], 'ms', d['value'], 'Mrays/s')"
done
