set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 400 python tools/two_stream_try.py > $O/two_stream.log 2>&1; grep "img/s" $O/two_stream.log
timeout -k 10 400 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -12 $O/t.log
