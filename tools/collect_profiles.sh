#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 stats + PMC traffic passes + bench JSONs into gpurun_out/$1
set -o pipefail
out=gpurun_out/${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --slices 1 > $out/bench_under_rocprof.json 2> $out/err.txt
echo "stats done"
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --slices 1 > /dev/null 2>&1; echo "fetch done"
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --slices 1 > /dev/null 2>&1; echo "write done"
timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --slices 1 > /dev/null 2>&1; echo "l2 done"
python bench.py --steps 5 --warmup 2 > $out/bench.json 2>/dev/null; cut -c1-200 $out/bench.json
for s in ico2 4boxes; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --scene $s > $out/bench_$s.json 2>/dev/null; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fix-row-index > $out/bench_fix_row_index.json 2>/dev/null
