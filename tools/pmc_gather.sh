#!/bin/bash
# TA / TCP / TD counters of the gather microbenchmark's kernels (tools/micro/gather_bench): what the vector memory pipe looks like when a gather saturates it
out=gpurun_out/${1:-pmcgather}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 rocprofv3 --pmc "${@:2}" --output-format csv -d $out/$1 -- tools/micro/gather_bench 48000 > $out/run_$1.txt 2> $out/err_$1.txt; echo "$1 $?"; }
run t1 TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_BUSY_CYCLES
run t2 TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum SQ_INSTS_VMEM_RD
run t3 TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(acc):
    d = {c: acc[k][c] / max(len(n[(k, c)]), 1) for c in acc[k]}
    busy = d.get("SQ_BUSY_CYCLES", 0) / 32
    print(k)
    print("   busy cycles/launch %.4g  TA busy %.3f  TD busy %.3f  cache accesses per CU-cycle %.3f  accesses per VMEM instr %.1f  L1->L2 reads per cache access %.3f  tag-conflict stall %.3f pending stall %.3f" % (
        busy, d.get("TA_TA_BUSY_sum", 0) / 256 / busy, d.get("TD_TD_BUSY_sum", 0) / 256 / busy, d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 256 / busy,
        d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / max(d.get("SQ_INSTS_VMEM_RD", 1), 1), d.get("TCP_TCC_READ_REQ_sum", 0) / max(d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1), 1),
        d.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0) / 256 / busy, d.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / busy))
PY
