for m in 3 4; do for i in 1 2; do IX_GEMM_MODE=$m timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $m frames/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), round(d['roofline']['kernel_ms_per_step'],1))"; done; done
