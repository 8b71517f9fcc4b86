#!/bin/bash
# quick look on the GPU box: parity subset, kernel stats of two 1-slice frames, the 3-slice frame time.  usage: quick_ab.sh <tag> [pytest -k expr]
out=gpurun_out/${1:-quick}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q -m gpu -k "${2:-golden or 1080p or film_bit_exact or dropin_loop or fused or closest_hit}" > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $out/pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --slices 1 > $out/bench_1slice.json 2> $out/err.txt; echo "stats $?"
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$out/stats/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.4:
        print("%-58s calls %4s avg %8.1f us  %5.1f%%" % (r["Name"].replace("mi355rt::", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $out/bench.json 2>> $out/err.txt
python3 -c "
import json; d=json.load(open('$out/bench.json')); print('frame', d['ms_per_step'], 'ms', d['value'], 'Mrays/s | other', d['other_semantics'])"
timeout -k 10 200 python bench.py --mode dropin --no-cpu-baseline --no-pmc > $out/dropin.json 2>> $out/err.txt
python3 -c "
import json; d=json.load(open('$out/dropin.json')); print('dropin', d['ms_per_step'], 'ms/step kernel', d['roofline']['avg_launch_ms'])"
