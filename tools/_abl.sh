for pf in 0 1 2; do EGT_BWD_PF=$pf timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('pf',$pf,round(d['ms_per_step'],3),round(d['roofline']['kernels']['k_block_bwd']['avg_us'],1))"; done
