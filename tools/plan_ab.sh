#!/bin/bash
# GPU: bench.py over builds (AB_LIBS) x chunk plans (AB_PLANS), 300 steps each.
for L in ${AB_LIBS:-libwdf_hip.so}; do for plan in ${AB_PLANS:-32,192,32}; do
  WDF_HIP_LIB=$PWD/differentiable-wdfs_amd/lib/wdf_hip/$L python bench.py --steps 300 --warmup 20 --plan $plan --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); tp=d['config']['time_parallel']; print('$L plan $plan', round(d['ms_per_step'],4), round(d['value']/1e9,1), tp['fwd_chunks'], tp['verify_status'])"
done; done
