#!/bin/bash
# usage: isa.sh <tag> [extra hipcc flags]  -> /tmp/probe/<tag>.s, resource usage of the fused kernel, stats of the FAST owned loop
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I /root/repo/differentiable-wdfs_amd/csrc --offload-device-only -S fused_isa.hip -o $tag.s -Rpass-analysis=kernel-resource-usage "$@" 2> $tag.rem
grep -A12 "Function Name: _ZN3wdf23clipper_fused" $tag.rem | grep -E "VGPRs:|AGPRs|Spill|Occupancy|ScratchSize" | sed 's/.*remark: *//'
python3 - $tag.s <<'PY'
import sys,re,collections
L=open(sys.argv[1]).read().split('\n')
loops=[i for i,l in enumerate(L) if 'Inner Loop Header' in l]
for n,li in enumerate(loops):
    end=loops[n+1] if n+1<len(loops) else len(L)
    body=[]
    for j in range(li,end):
        body.append(L[j])
        if 's_waitcnt vmcnt(8)' in L[j] and sum(1 for b in body if 'buffer_store_dwordx2' in b and ' nt' in b)==8: break
    else:
        continue
    if any('s_cbranch_execnz' in b for b in body): continue
    ins=[b.split()[0] for b in body if b.startswith('\t') and not b.strip().startswith(';') and not b.strip().startswith('.')]
    c=collections.Counter(re.sub(r'_e(32|64)$','',i) for i in ins)
    pk=sum(v for k,v in c.items() if k.startswith('v_pk_'))
    tr=sum(v for k,v in c.items() if k in('v_exp_f32','v_log_f32','v_rcp_f32'))
    va=sum(v for k,v in c.items() if k.startswith('v_'))-pk-tr
    nop=sum(1+int(b.split()[1]) for b in body if b.strip().startswith('s_nop'))
    print(f"FAST owned loop @line {li}: {len(ins)} instr per tile of 8 steps: pk {pk} plain {va} trans {tr} s_nop-cycles {nop} movs {c.get('v_mov_b32',0)+c.get('v_mov_b64',0)} scratch {sum(1 for b in body if 'scratch_' in b)} branches {sum(1 for b in body if 's_cbranch' in b)}; issue-cycle model/step {(pk*4+va*2+tr*8+nop)/8:.0f}")
PY
