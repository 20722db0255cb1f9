#!/bin/bash
# knob_sweep.sh — run on the GPU box: Gram chunk-rows / slices sweep at C4 and sweep chain-length / grid sweep at C3 (environment knobs
# read once at partls_create).  Output: gpurun_out/knobs.log
set -o pipefail
O=gpurun_out/knobs.log; mkdir -p gpurun_out; : > $O
for cr in 0 32 64 128 448; do for S in 0 8 30; do
  echo "GRAM cr=$cr S=$S" >> $O; PARTLS_GRAM_CR=$cr PARTLS_GRAM_S=$S timeout -k 10 120 python3 tools/gram_bench.py 1000000 512 16 6 >> $O 2>&1 || exit 1
done; done
for cl in 128 256 512 1024; do for g in 0; do
  echo "SWEEP chain_len=$cl" >> $O; PARTLS_CHAIN_LEN=$cl timeout -k 10 200 python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline >> $O 2>&1 || exit 1
done; done
