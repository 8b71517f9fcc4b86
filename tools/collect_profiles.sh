#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats of the bench command + the bench JSON lines (which carry
# their own rocprofv3 --pmc child passes) into gpurun_out/$1.  tools/publish_profiles.sh copies the summaries to profiles/.
set -o pipefail
out=gpurun_out/${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $out
# 1. kernel stats of the headline command (one slice, so that per-kernel durations mean something)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-groups --no-dropin > $out/bench_under_rocprof.json 2> $out/err_stats.txt; echo "stats done $?"
# 2. kernel trace of the drop-in loop (timeline: are there host-sync gaps between the launches of a step?)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dropin_trace -- python3 bench.py --mode dropin --steps 300 --warmup 20 --no-cpu-baseline --no-pmc > $out/dropin_under_rocprof.json 2> $out/err_dropin.txt; echo "dropin trace done $?"
python3 tools/dropin_timeline.py $out/dropin_trace > $out/dropin_timeline.txt 2>&1; tail -5 $out/dropin_timeline.txt
# 3. the bench lines themselves
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/err_bench.txt; echo "bench $?"; cut -c1-300 $out/bench.json
timeout -k 10 300 python bench.py --mode dropin > $out/dropin.json 2> $out/err_dropin2.txt; echo "dropin $?"; cut -c1-300 $out/dropin.json
for s in ico2 4boxes; do timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-groups --no-dropin --scene $s > $out/bench_$s.json 2>/dev/null; echo "$s $?"; done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-groups --no-dropin --fix-row-index > $out/bench_fix_row_index.json 2>/dev/null; echo "fix $?"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-groups --no-dropin --true-closest-hit > $out/bench_true_closest_hit.json 2>/dev/null; echo "tch $?"
# 4. N > 1 rehearsal (2 ranks share the GPU over gloo; bench.py starts them itself) and the strong-scaling shares
MI355RT_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $out/bench_2ranks_shared_gpu.json 2> $out/err_2ranks.txt; echo "2 ranks $?"
timeout -k 10 300 python tools/weak_scaling_probe.py > $out/scaling_probe_1gpu.txt 2>&1; echo "probe $?"; tail -4 $out/scaling_probe_1gpu.txt
# 5. counters of every kernel (vector memory pipe) and of the gather microbenchmark
tools/pmc_tcp.sh ${1:-r03}/pmctcp _kernel > /dev/null 2>&1; cp gpurun_out/${1:-r03}/pmctcp/summary.txt $out/pmc_tcp_all_kernels.txt
tools/pmc_gather.sh ${1:-r03}/pmcgather > $out/pmc_gather_microbench.txt 2>&1
tools/micro/gather_bench 48000 > $out/gather_bench.txt 2>&1
