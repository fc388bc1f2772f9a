#!/bin/bash
# A/B the variant libraries built by tools/build_variant.sh on the GPU box: bench value + dominant kernel time per variant.
# Usage (inside gpurun): tools/ab.sh "<bench args>" name1 name2 ...   (name "default" = the in-tree library)
args=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset EGT_AMD_LIB; else export EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_$v.so; fi
  python bench.py $args --no-cpu-baseline 2> gpurun_out/ab_$v.err | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline'] or {}
        k=r.get('kernels') or {}
        print('$v', round(d['value']), 'graphs/s', round(d['ms_per_step'],3), 'ms', {n:round(x['avg_us'],1) for n,x in list(k.items())[:3]})
"
  grep -A14 "phase cycles" gpurun_out/ab_$v.err
done
