#!/bin/bash
# concurrency analysis of the default bench (2 groups): in-situ kernel time per queue vs wall
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ov; rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" > /tmp/ov.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/ov/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
ends=[i for i,r in enumerate(rows) if 'clip_adam' in r['Kernel_Name']]
# timed region: 12 group steps (60 steps / 5) = last 12 clip_adam before the single-chain tail (13 single steps)
a,b=ends[-13-12-1],ends[-13-1]
seg=rows[a+1:b+1]
t0=int(seg[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in seg)
wall=(t1-t0)/1e6
tot=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e6
# union busy
ev=sorted([(int(r['Start_Timestamp']),1) for r in seg]+[(int(r['End_Timestamp']),-1) for r in seg])
depth=0; last=t0; hist=collections.Counter()
for t,d in ev:
    hist[depth]+=t-last; last=t; depth+=d
print('wall %.2f ms, sum of kernel durations %.2f ms (%.2fx), kernels %d'%(wall,tot,tot/wall,len(seg)))
for k in sorted(hist): print('  depth %d: %.2f ms (%.1f%%)'%(k,hist[k]/1e6,100*hist[k]/(t1-t0)))
q=collections.Counter(r['Queue_Id'] for r in seg); print('queues',dict(q))
PY
