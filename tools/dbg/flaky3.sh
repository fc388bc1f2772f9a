mkdir -p gpurun_out/flaky; n=0
for i in $(seq 1 ${NREP:-20}); do
  EGT_BWD_TL=5 timeout 300 python -m pytest -q -m gpu -p no:cacheprovider ${SEL} > gpurun_out/flaky/r.log 2>&1 || { n=$((n+1)); grep "AssertionError: \|^FAILED" gpurun_out/flaky/r.log | head -4; }
done
echo "failures: $n of $NREP"
