for K in 8 12 16 20 24 32; do
  python bench.py --no-cpu-baseline --steps 40 --plan $K,160,32 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); tp=d['config']['time_parallel']; print('K', tp['fwd_chunks'], 'G/s %.1f'%(d['value']/1e9), 'step %.4f'%d['ms_per_step'], 'kernel_ms', {k: round(v['median'], 4) for k, v in d['kernel_ms'].items()}, tp['warm_start'])"
done
for KB in 16 64; do
  python bench.py --no-cpu-baseline --steps 40 --plan 16,160,$KB 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); tp=d['config']['time_parallel']; print('KB', tp['bwd_chunks'], 'G/s %.1f'%(d['value']/1e9), 'step %.4f'%d['ms_per_step'], 'kernel_ms', {k: round(v['median'], 4) for k, v in d['kernel_ms'].items()})"
done
