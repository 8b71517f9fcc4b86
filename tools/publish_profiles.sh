#!/bin/bash
# Copies the summaries of gpurun_out/$1 into profiles/ (tracked), prefix $2
src=gpurun_out/${1:-r03}; pre=profiles/${2:-r03}
cp $(ls -t $src/stats/*/*_kernel_stats.csv | head -1) ${pre}_rocprofv3_kernel_stats_bench.csv
cp $(ls -t $src/dropin_trace/*/*_kernel_stats.csv | head -1) ${pre}_rocprofv3_kernel_stats_dropin.csv
cp $src/dropin_timeline.txt ${pre}_dropin_timeline.txt
cp $src/bench.json ${pre}_bench_n1_thai2.json
cp $src/bench_under_rocprof.json ${pre}_bench_n1_thai2_under_rocprofv3.json
cp $src/dropin.json ${pre}_bench_dropin_thai2_1024x768.json
cp $src/bench_ico2.json ${pre}_bench_n1_ico2.json; cp $src/bench_4boxes.json ${pre}_bench_n1_4boxes.json
cp $src/bench_fix_row_index.json ${pre}_bench_n1_thai2_fix_row_index.json
cp $src/bench_true_closest_hit.json ${pre}_bench_n1_thai2_true_closest_hit.json
cp $src/bench_2ranks_shared_gpu.json ${pre}_bench_2ranks_shared_gpu_rehearsal.json
cp $src/scaling_probe_1gpu.txt ${pre}_scaling_probe_1gpu.txt
cp $src/pmc_tcp_all_kernels.txt ${pre}_pmc_tcp_all_kernels.txt
cp $src/pmc_gather_microbench.txt ${pre}_pmc_gather_microbench.txt
cp $src/gather_bench.txt ${pre}_gather_bench.txt
