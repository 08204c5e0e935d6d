#!/bin/bash
# Instruction-cache behaviour per kernel inside a 3-model lock-step step (GPU box): SQC_ICACHE_* per dispatch, averaged per kernel name
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pic; rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d /tmp/pic -- python3 $R/bench.py --steps 36 --warmup 4 --workload c2 --no-cpu-baseline --timed-only --concurrent-folds 1 --fold-group ${GROUP:-3} > /tmp/pic.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pic/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('void tile_gemm_kernel','tgk')[:60]+' g%s'%r.get('Grid_Size','')
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']=='SQC_ICACHE_REQ': cnt[k]+=1
rows=[]
for k,v in agg.items():
    n=max(cnt[k],1); rows.append((v['SQC_ICACHE_MISSES']/n, k, n, v['SQC_ICACHE_REQ']/n, v['SQC_ICACHE_HITS']/n, v['SQC_ICACHE_MISSES_DUPLICATE']/n))
print('%-72s %6s %10s %10s %10s %10s'%('kernel','n','req','hits','misses','dup'))
for m,k,n,rq,h,d in sorted(rows,reverse=True)[:400]:
    print('%-72s %6d %10.0f %10.0f %10.0f %10.0f'%(k,n,rq,h,m,d))
PY
