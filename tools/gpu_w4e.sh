# F(4x4) in the training path: layer / train parity tests, then the train bench with the F(4x4) rule on and off on the same box.  -> gpurun_out/w4e/
cd $GRAFT_REPO_ROOT
O=gpurun_out/w4e; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_layers_gpu.py tests/test_amp_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 300 python bench.py --mode train > $O/bench_train_w4.json 2> $O/bench_train_w4.err; cut -c1-250 $O/bench_train_w4.json
FD_WINOGRAD4=0 timeout -k 10 300 python bench.py --mode train > $O/bench_train_w2.json 2> $O/bench_train_w2.err; cut -c1-250 $O/bench_train_w2.json
