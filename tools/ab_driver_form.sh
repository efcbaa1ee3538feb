for rep in 1 2 3; do for L in ${AB_LIBS:-libwdf_prev.so libwdf_hip.so}; do
  WDF_HIP_LIB=$PWD/differentiable-wdfs_amd/lib/wdf_hip/$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L driver form', round(d['ms_per_step'],4), round(d['value']/1e9,1), d['kernel_ms']['fused_step'])"
done; done
for L in ${AB_LIBS:-libwdf_prev.so libwdf_hip.so}; do
  WDF_HIP_LIB=$PWD/differentiable-wdfs_amd/lib/wdf_hip/$L python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-cold --no-batch-major --no-strong-proxy 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L 300 steps', round(d['ms_per_step'],4), round(d['value']/1e9,1))"
done
python tools/warm_trace.py 12 2>&1 | tail -2
