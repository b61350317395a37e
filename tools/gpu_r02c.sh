set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 120 python tools/grad_strides.py > $O/strides.log 2>&1; tail -12 $O/strides.log
timeout -k 10 300 python tools/two_stream_try.py > $O/two_stream.log 2>&1; cat $O/two_stream.log | grep -v amdgpu.ids
timeout -k 10 300 python -m pytest tests/test_train_gpu.py tests/test_effnet_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -5 $O/t.log
