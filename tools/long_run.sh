#!/bin/bash
# GPU: the bench's training loop at several lengths (sustained rate vs the short runs)
for n in ${STEPS:-3000 30000}; do
  python bench.py --steps $n --warmup 20 --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps $n', round(d['ms_per_step'],4), round(d['value']/1e9,1), d['kernel_ms']['fused_step'])"
done
