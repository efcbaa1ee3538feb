#!/bin/bash
# usage: tools/asmstat.sh <mangled-kernel-prefix>   -- rebuild, dump the kernel's ISA to /tmp/k.s, print stats
set -e
CS="/root/repo/differentiable-wdfs_amd/csrc"
make -C "$CS" 2>&1 | grep -E "error|warning" | head -20 || true
make -C "$CS" asm > /dev/null 2>&1
S="$CS/build/wdf_capi-hip-amdgcn-amd-amdhsa-gfx950.s"
K="$1"
a=$(grep -n "^${K}.*:" "$S" | head -1 | cut -d: -f1)
b=$(grep -n "amdhsa_kernel ${K}" "$S" | head -1 | cut -d: -f1)
sed -n "${a},${b}p" "$S" > /tmp/k.s
echo "instructions: $(grep -cE '^\s+(v_|s_|global_|buffer_|ds_|flat_)' /tmp/k.s)"
grep -A9 "Function Name: ${K}" "$CS/build/resource_usage.txt" | grep -E "VGPRs:|SGPRs:|Occupancy|Scratch" | sed 's/.*remark: [^ ]* *//'
