#!/bin/bash
# single-chain per-kernel profile of one training step (GPU box): prints the top kernels by time per step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pstep; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pstep -- python3 $R/bench.py --steps ${STEPS:-20} --warmup 4 --workload c2 --no-cpu-baseline --timed-only --concurrent-folds 1 --fold-group ${GROUP:-1} > /tmp/pstep.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pstep/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
ends=[i for i,r in enumerate(rows) if 'clip_adam' in r['Kernel_Name']]
skip=int(__import__('os').environ.get('SKIP','0'))
a,b=ends[-6-skip],ends[-2-skip]; seg=rows[a+1:b+1]; n=4
agg=collections.defaultdict(lambda:[0,0])
for r in seg:
    k=r['Kernel_Name'].replace('void tile_gemm_kernel','tgk')[:44]+' g%dx%dx%d'%(int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])),int(r['Grid_Size_Y']),int(r['Grid_Size_Z'])); agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot=sum(v[1] for v in agg.values())
print('step busy %.3f ms, %d kernels'%(tot/n/1e6, len(seg)//n))
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(__import__('os').environ.get('TOPN','60'))]:
    print('%-56s n=%5.1f tot %.3f ms avg %6.1f us'%(k,c/n,t/n/1e6,t/c/1e3))
PY
