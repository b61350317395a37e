# GEMM-addressed loader variant of the direct kernel (FD_CONV_GEMM=1, default) against the generic loader (=0): parity tests, bench, per-layer times
cd $GRAFT_REPO_ROOT
O=gpurun_out/gemm; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_layers_gpu.py tests/test_model_gpu.py tests/test_trunk_dump.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times_gemm1.tsv > /dev/null 2>&1
FD_CONV_GEMM=0 timeout -k 10 300 python bench.py --inflight 1 --layer-times $O/layer_times_gemm0.tsv > /dev/null 2>&1
tail -1 $O/layer_times_gemm1.tsv; tail -1 $O/layer_times_gemm0.tsv
timeout -k 10 400 python bench.py --no-train-step --no-cpu-baseline --no-fast-mode > $O/bench_gemm1.json 2> $O/bench_gemm1.err; cut -c1-200 $O/bench_gemm1.json
FD_CONV_GEMM=0 timeout -k 10 400 python bench.py --no-train-step --no-cpu-baseline --no-fast-mode > $O/bench_gemm0.json 2> $O/bench_gemm0.err; cut -c1-200 $O/bench_gemm0.json
