#!/bin/bash
# Long parity soak on the GPU box: tools/parity_fuzz.py over seeds $2..$3, 300 cases each, the wider modes cycling with the seed.  usage: soak.sh <outdir> <first seed> <last seed>
out=gpurun_out/$1; mkdir -p $out
for s in $(seq $2 $3); do
  case $((s % 4)) in
    0) m="";;
    1) m="FUZZ_SPP=1 FUZZ_SOUP=1";;
    2) m="FUZZ_BUILD=1 FUZZ_SOUP=1";;
    3) m="FUZZ_WILD=1 FUZZ_SPP=1 FUZZ_SOUP=1 FUZZ_BUILD=1";;
  esac
  env $m timeout -k 10 400 python tools/parity_fuzz.py 300 $s > $out/fuzz$s.log 2>&1 || { echo "seed $s ($m) FAILED"; tail -5 $out/fuzz$s.log; exit 1; }
  echo "seed $s ($m): $(tail -1 $out/fuzz$s.log)"
done
