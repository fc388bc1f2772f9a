#!/bin/bash
# round 5, after the last kernel change: randomised parity sweeps + determinism soak (the round's start-up staging / partial-gather
# changes touch every geometry of the pair kernels)
out=gpurun_out/r05_sweeps; mkdir -p $out
log=$out/r05_final_sweep_soak.log; : > $log
for s in 2024 5 9 17 23; do
  echo "== sweep_parity seed $s" >> $log
  EGT_SWEEP_SEED=$s timeout 900 python tools/sweep_parity.py 2>&1 | grep -E "^sweep|FAIL" >> $log
done
echo "== sweep_de8 12" >> $log
timeout 900 python tools/sweep_de8.py 12 2>&1 | grep -E "setting|sweep|FAIL" >> $log
echo "== sweep_de8 20, seed 77" >> $log
EGT_SWEEP_SEED=77 timeout 900 python tools/sweep_de8.py 20 2>&1 | grep -E "setting|sweep|FAIL" >> $log
echo "== sweep_mfma" >> $log
timeout 900 python tools/sweep_mfma.py 7 60 2>&1 | grep -E "sweep_mfma|FAIL" >> $log
echo "== soak_determinism 20" >> $log
timeout 900 python tools/soak_determinism.py 20 2>&1 | tail -4 >> $log
echo "== EGT_BWD_V7=1 / EGT_BWD_MATMUL=bf16x3 suites" >> $log
timeout 600 python -m pytest tests/test_bwd_v7_gpu.py tests/test_bwd_modes_gpu.py -q -m gpu 2>&1 | tail -1 >> $log
cat $log
