# F(4x4) integration check: model / layer parity tests, then the bench with the F(4x4) rule on and off on the same box.  -> gpurun_out/w4b/
cd $GRAFT_REPO_ROOT
O=gpurun_out/w4b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_layers_gpu.py tests/test_trunk_dump.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 400 python bench.py --no-train-step --no-cpu-baseline > $O/bench_w4.json 2> $O/bench_w4.err; tail -2 $O/bench_w4.err; cut -c1-300 $O/bench_w4.json
FD_WINOGRAD4=0 timeout -k 10 400 python bench.py --no-train-step --no-cpu-baseline --no-fast-mode > $O/bench_w2.json 2> $O/bench_w2.err; cut -c1-300 $O/bench_w2.json
