#!/bin/bash
# A/B of library builds on ONE box, interleaved (A B A B ...): rank 0's share of the strong-scaled headline frame (N = 8) and the N = 1 frame.
# usage: ab.sh <rounds> libA.so libB.so ...   (paths relative to raytracer-rs_amd/; extra env through the caller's environment)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    echo -n "$lib: "; MI355RT_LIB=$GRAFT_REPO_ROOT/raytracer-rs_amd/$lib python3 tools/share_sweep.py 2>&1 | tail -1
  done
done
