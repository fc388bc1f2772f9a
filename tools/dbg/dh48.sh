mkdir -p gpurun_out/dh48
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_capi_graph_gpu.py tests/test_graph_gpu.py tests/test_narrow_gpu.py -m gpu -q -x -k "not sweep and not random_de8" 2>&1 | tail -15
bash tools/dbg/ab_lib.sh pre "zinc100k_n37 zinc500k_n64"
