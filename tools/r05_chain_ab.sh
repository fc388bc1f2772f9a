#!/bin/bash
# round 5: the node-side MFMA chains of the backward prologue / forward epilogue on several accumulators, against the build before ("prechain")
out=gpurun_out/r05_chain; mkdir -p $out
timeout 1500 python -m pytest tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_narrow_gpu.py tests/test_block_variants_gpu.py tests/test_bwd_v7_gpu.py tests/test_graph_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.txt
for i in 1 2; do
  tools/ab.sh "--no-graph-leg --graph off --steps 40" default prechain 2>&1 | grep graphs | tee -a $out/ab.txt
done
for wl in cifar10_n150 pattern500k_n120 zinc100k_n37; do
  tools/ab.sh "--workload $wl --no-graph-leg --steps 30" default prechain 2>&1 | grep graphs | sed "s/^/$wl /" | tee -a $out/ab.txt
done
