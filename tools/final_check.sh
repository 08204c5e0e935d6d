# end-of-round check on the GPU box: smoke, default bench line, config-5 step, 2-rank rehearsals (gloo, both ranks on the one GPU)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > $O/bench_default.json; cat $O/bench_default.json
python bench.py --workload c5 2>/dev/null | tail -1 > $O/c5.json; cat $O/c5.json
bash tools/prof_c5.sh > $O/c5_kernels.txt 2>&1; head -3 $O/c5_kernels.txt
MMS_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_2rank_fold.json; cat $O/bench_2rank_fold.json
MMS_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 30 --warmup 5 --mode ddp --global-cox 2>/dev/null | tail -1 > $O/bench_2rank_ddp.json; cat $O/bench_2rank_ddp.json
