mkdir -p gpurun_out/nrwm
timeout 1500 python -m pytest tests/test_narrow_gpu.py tests/test_block_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/nrwm/pytest2.log 2>&1; tail -3 gpurun_out/nrwm/pytest2.log
EGT_SWEEP_SEED=5 timeout 600 python tools/sweep_parity.py 2>&1 | tail -2
BS="96 102 104 128 152 154 160" bash tools/dbg/bsweep_narrow.sh
WL=pattern500k_n120_b128 BS="16 32 64 96 128 192" bash tools/dbg/bsweep_narrow.sh
