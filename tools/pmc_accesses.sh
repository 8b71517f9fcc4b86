#!/bin/bash
# cache-line accesses of the trace kernels of ONE N=1 frame under environment variants (one rocprofv3 --pmc pass each): pmc_accesses.sh VAR=a,b,c
out=gpurun_out/pmcacc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
var=${1%%=*}; vals=${1#*=}
for v in ${vals//,/ }; do
  rm -rf $out/p
  env $var=$v timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $out/p -- python3 tools/share_sweep.py --worlds=1 > $out/run.txt 2> $out/err.txt
  python3 - "$var=$v" <<PY
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$out/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_kernel" not in k: continue
        kind = "primary" if "<true" in k else "secondary"
        acc[kind][r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[1], "| " + open("$out/run.txt").read().strip().split("|")[-1].strip(), "|", "  ".join("%s: %.3g accesses, %.3g vmem rd, busy %.3g" % (k, d["TCP_TOTAL_CACHE_ACCESSES_sum"], d["SQ_INSTS_VMEM_RD"], d["SQ_BUSY_CYCLES"] / 32) for k, d in sorted(acc.items())))
PY
done
