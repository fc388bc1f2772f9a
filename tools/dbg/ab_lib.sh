# A/B of two builds on the same box:  bash tools/dbg/ab_lib.sh <variant-name> "<workloads>"   (variant = egt_amd/lib/var/libegt_<name>.so)
V=$1; shift
for rep in 1 2; do for wl in $1; do for L in new $V; do
  if [ $L = new ]; then unset EGT_AMD_LIB; else export EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_$L.so; fi
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-graph-leg --graph off --steps 12 --warmup 3 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$wl $L', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}' for n, v in list(k.items())[:5]))
"
done; done; done
