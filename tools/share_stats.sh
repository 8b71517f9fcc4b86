#!/bin/bash
# kernel stats of rank 0's share of a weak-scaling frame for N = 1 and N = 8 (1 slice each, so that kernel times mean something)
out=gpurun_out/${1:-share}; mkdir -p $out; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 1 8; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/w$w -- python3 tools/share_stats.py $w 1 > $out/w$w.txt 2> $out/err$w.txt; cat $out/w$w.txt
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$out/w$w/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.5:
        print("   %-56s calls %4s avg %8.1f us total %8.2f ms" % (r["Name"].replace("mi355rt::", "")[:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
