set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
FD_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 6 --warmup 2 --no-fast-mode --no-train-step > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo rc=$?; tail -5 $O/bench_2rank_gloo.err | cut -c1-300; cat $O/bench_2rank_gloo.json | cut -c1-700
FD_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --mode train --steps 3 --warmup 1 > $O/bench_2rank_train.json 2> $O/bench_2rank_train.err; echo rc=$?; tail -3 $O/bench_2rank_train.err | cut -c1-300; cat $O/bench_2rank_train.json | cut -c1-400
timeout -k 10 200 python -m pytest tests/test_postproc_gpu.py -q -m gpu 2>&1 | tail -2
python bench.py --no-fast-mode --no-cpu-baseline --no-train-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['postproc']['batch16']['decode_us'], d['postproc']['batch16']['decode_hbm_frac'])"
