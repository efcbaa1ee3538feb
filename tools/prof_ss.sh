cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_ss
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ss -o p -- python tools/ss_probe.py > gpurun_out/r03_ss_probe.txt 2>&1
cp $(find gpurun_out/prof_ss -name "p_kernel_stats.csv" | head -1) gpurun_out/r03_ss_kernel_stats.csv
grep -v "amdgpu.ids\|UserWarning\|Consider using\|rootp = " gpurun_out/r03_ss_probe.txt
