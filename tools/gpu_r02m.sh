cd $GRAFT_REPO_ROOT
O=gpurun_out/r02mn; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_mnfcos_gpu.py -q -m gpu > $O/t.log 2>&1; tail -30 $O/t.log
timeout -k 10 300 python bench.py --model MNFCOS --no-fast-mode --no-train-step --no-cpu-baseline > $O/bench_mnfcos.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-400 $O/bench_mnfcos.json
