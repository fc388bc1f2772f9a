#!/bin/bash
# round 5: start-up loads batched into one round trip, same box:
#   prev2   before (load -> wait -> LDS store loops)
#   stage   forward K/V staging + backward row staging batched
#   default stage + the prologue's dQ / dK / dV partial gather in one round (three waits -> one)
out=gpurun_out/r05_start; mkdir -p $out
tools/ab.sh "--no-graph-leg --graph off --steps 50" default stage prev2 default stage prev2 2>&1 | tee $out/ab_headline.txt
for wl in cifar10_n150 pattern500k_n120_b128 pattern500k_n120 zinc100k_n37; do
  tools/ab.sh "--workload $wl --no-graph-leg --graph off --steps 30" default stage prev2 default stage prev2 2>&1 | sed "s/^/$wl /" | tee -a $out/ab_others.txt
done
unset EGT_AMD_LIB
timeout 1500 python -m pytest tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_block_variants_gpu.py tests/test_narrow_gpu.py tests/test_bwd_v7_gpu.py tests/test_bwd_modes_gpu.py -x -q -m gpu 2>&1 | tail -3
