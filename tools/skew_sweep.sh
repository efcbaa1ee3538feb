#!/bin/bash
# GPU: the one-pass step over the skew of its chunk spans (older wave of a SIMD pair owns L + skew steps, the younger L - skew).
for rep in 1 2; do for sk in 0 16 32 48 64; do
  WDF_FUSED_SKEW_STEPS=$sk WDF_MAX_WARM_TILES=${MWT:-4} python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('skew $sk', round(d['ms_per_step'],4), round(d['value']/1e9,1), d['kernel_ms']['fused_step']['median'], d['config']['time_parallel']['verify_status']['n_bad'])"
done; done
