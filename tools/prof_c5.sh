#!/bin/bash
# per-kernel profile of the config-5 training step (GPU box): average duration of every kernel of the step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pc5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -- python3 $R/tools/bench_c5.py --steps ${STEPS:-50} --batch ${BATCH:-2048} > /tmp/pc5.log 2>&1
tail -1 /tmp/pc5.log
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pc5/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
ends=[i for i,r in enumerate(rows) if 'clip_adam' in r['Kernel_Name']]
a,b=ends[-22],ends[-2]; seg=rows[a+1:b+1]; n=20
agg=collections.OrderedDict()
for r in seg:
    k=r['Kernel_Name'].replace('void tile_gemm_kernel','tgk').replace('(anonymous namespace)::','')[:60]+' g%dx%dx%d'%(int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])),int(r['Grid_Size_Y']),int(r['Grid_Size_Z']))
    agg.setdefault(k,[0,0]); agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot=sum(v[1] for v in agg.values())
print('step busy %.3f ms, wall %.3f ms, %d kernels'%(tot/n/1e6,(int(rows[b]['End_Timestamp'])-int(rows[a]['End_Timestamp']))/n/1e6,len(seg)//n))
for k,(c,t) in agg.items():
    print('%-76s n=%4.1f avg %7.1f us'%(k,c/n,t/c/1e3))
PY
