# usage: bash tools/ab_libs.sh "<bench args>" lib1.so lib2.so ...   (alternates the builds, two rounds)
ARGS="$1"; shift
for round in 1 2; do for L in "$@"; do
  WDF_HIP_LIB=$PWD/differentiable-wdfs_amd/lib/wdf_hip/$L python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); tp=d['config']['time_parallel']; print('$L', 'G/s %.1f'%(d['value']/1e9), 'step %.4f'%d['ms_per_step'], 'kernel_ms', {k: round(v['median'], 4) for k, v in d['kernel_ms'].items()})"
done; done
