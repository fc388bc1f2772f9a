#!/bin/bash
# round 5: k_block_bwd_v5 start-up -- first e tile / K / V of a wave and the prologue's loads requested at kernel entry (one round trip
# for the whole pre-loop phase) -- against the build before ("preearly"), same box
out=gpurun_out/r05_early; mkdir -p $out
timeout 1500 python -m pytest tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_block_variants_gpu.py tests/test_bwd_v7_gpu.py tests/test_bwd_modes_gpu.py tests/test_graph_gpu.py tests/test_sweep_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.txt
for i in 1 2; do
  tools/ab.sh "--no-graph-leg --graph off --steps 40" default preearly 2>&1 | grep graphs | tee -a $out/ab.txt
done
for wl in zinc100k_n37 zinc500k_n64_full; do
  tools/ab.sh "--workload $wl --no-graph-leg --steps 30" default preearly 2>&1 | grep graphs | sed "s/^/$wl /" | tee -a $out/ab.txt
done
