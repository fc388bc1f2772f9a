#!/bin/bash
# round 5: fragment-major prepared Wqkv / Wo for the node-side epilogue / prologue of the pair kernels, against the build before ("prefrag")
out=gpurun_out/r05_frag; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest.txt
for i in 1 2; do
  tools/ab.sh "--no-graph-leg --graph off --steps 40" default prefrag 2>&1 | grep graphs | tee -a $out/ab.txt
done
for wl in zinc100k_n37 cifar10_n150 pattern500k_n120 pattern500k_n120_b128; do
  tools/ab.sh "--workload $wl --no-graph-leg --steps 30" default prefrag 2>&1 | grep graphs | sed "s/^/$wl /" | tee -a $out/ab.txt
done
