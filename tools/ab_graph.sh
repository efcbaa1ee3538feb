for rep in 1 2; do for g in off on; do
  python bench.py --steps 300 --warmup 20 --graph $g --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph $g 300 steps', round(d['ms_per_step'],4), round(d['value']/1e9,1), d.get('step_launch'), d['kernel_ms']['fused_step']['median'])"
  python bench.py --steps 20 --warmup 3 --graph $g --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph $g driver form', round(d['ms_per_step'],4), round(d['value']/1e9,1), d.get('step_launch'), d['kernel_ms']['fused_step']['median'])"
done; done
