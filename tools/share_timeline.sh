#!/bin/bash
# kernel timeline of rank 0's share of the strong-scaled headline frame (N = 8), last frame: tools/ktrace_summary.py
set -o pipefail
out=gpurun_out/${1:-share_tl}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/share_sweep.py --worlds=${2:-8} > $out/run.txt 2> $out/err.txt
cat $out/run.txt
python3 tools/ktrace_summary.py $out/trace ${3:-16}
