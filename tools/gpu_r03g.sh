cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_amp_gpu.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout -k 10 200 python tools/time_wgrad16.py 2>&1 | grep -v amdgpu | tee $O/time_wgrad16.txt
timeout -k 10 300 python bench.py --mode train --amp > $O/bench_train_amp.json 2> $O/err; echo "train amp rc=$?"; cut -c1-200 $O/bench_train_amp.json
