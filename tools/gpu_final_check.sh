cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
# N = 2 control flow on one GPU (gloo carries the collectives; never a measurement)
FD_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --no-train-step 2>/dev/null | tail -1 | cut -c1-400
FD_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --no-train-step 2>/dev/null | tail -1 | cut -c1-200
FD_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --mode train --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-mode --no-train-step 2>/dev/null | tail -1 | cut -c1-120
timeout -k 10 300 python -m pytest tests/test_layers_gpu.py tests/test_model_gpu.py -q -m gpu -k "groupnorm or gn or hisfcos or fcos" 2>&1 | tail -2
