#!/bin/bash
# GPU: the MLP-root training step against the forward's chunk count and the warm-start controller's first value.
# usage: tools/mlp_warm_sweep.sh [root=mlp2x16] ["k list"="0 16 24 32"] ["W list"="0"]      (0 = the planner's / controller's own)
root=${1:-mlp2x16}
for k in ${2:-0 16 24 32}; do for w in ${3:-0}; do
  env $([ $k != 0 ] && echo WDF_MLP_K_FWD=$k) $([ $w != 0 ] && echo WDF_MLP_WARM_W=$w) WDF_MLP_TRACE_WARMUP=1 \
    python bench.py --root $root --steps 200 --warmup 60 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); tp=d['config']['time_parallel']; tr=tp.get('fwd_warmup_trace',[])
print('k_fwd=%s start=$w  ms/step %.4f  kernels %s  warm-up in use %s (cold %s)  trace %s  status %s' % (tp['fwd_chunks'], d['ms_per_step'], {k:round(v['median'],4) for k,v in (d.get('kernel_ms') or {}).items()}, tp['fwd_warmup_steps_used'], tp['fwd_warmup_steps_cold'], [tr[i] for i in range(0,len(tr),20)], tp['verify_status']))"
done; done
