#!/bin/bash
# Copies the summaries of gpurun_out/$1 into profiles/ (tracked), prefix $2
src=gpurun_out/${1:-r01}; pre=profiles/${2:-r01}
cp $(ls -t $src/stats/*/*_kernel_stats.csv | head -1) ${pre}_rocprofv3_kernel_stats_bench.csv
cp $src/bench.json ${pre}_bench_n1_thai2.json
cp $src/bench_under_rocprof.json ${pre}_bench_n1_thai2_under_rocprofv3.json
cp $src/bench_ico2.json ${pre}_bench_n1_ico2.json; cp $src/bench_4boxes.json ${pre}_bench_n1_4boxes.json
cp $src/bench_fix_row_index.json ${pre}_bench_n1_thai2_fix_row_index.json
{ echo "# rocprofv3 --pmc passes over: python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (one 64-spp thai2 1080p frame + one 4-spp instrumented frame)"
  echo "# separate passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum ; values summed over the listed dispatches; FETCH/WRITE_SIZE in KB"
  for x in pmc_fetch pmc_write pmc_l2; do echo "== $x"; python tools/pmc_summary.py $src/$x _kernel; done; } > ${pre}_pmc_traffic_summary.txt
