mkdir -p gpurun_out/occ
for cfg in ${CFGS:-"cifar10_n150 128" "pattern500k_n120 128" "pattern500k_n120 16"}; do set -- $cfg
  echo "== $1 B=$2" 
  EGT_AMD_LIB=egt_amd/lib/var/libegt_nrwtime.so EGT_BENCH_B=$2 timeout 300 python bench.py --workload $1 --no-cpu-baseline --no-graph-leg --no-prof --steps 10 --warmup 3 2>&1 >/dev/null | grep -A9 "section cycles"
done 2>&1 | tee gpurun_out/occ/nrw_time.log
