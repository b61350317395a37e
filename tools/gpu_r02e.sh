set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_postproc_gpu.py tests/test_model_gpu.py tests/test_layers_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -8 $O/t.log
timeout -k 10 300 python bench.py --inflight 1 --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_if1.json 2> $O/bench_if1.err; cat $O/bench_if1.json; tail -3 $O/bench_if1.err
timeout -k 10 300 python bench.py --inflight 2 --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_if2.json 2> $O/bench_if2.err; cat $O/bench_if2.json; tail -3 $O/bench_if2.err
