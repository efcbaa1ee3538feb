cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_a -o p -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --plan 16,160,32 > gpurun_out/prof_r02_a.log 2>&1
python tools/trace_gaps.py gpurun_out/prof_r02_a/p_kernel_trace.csv 200 > gpurun_out/prof_r02_a_gaps.txt 2>&1
