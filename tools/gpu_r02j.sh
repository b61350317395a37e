set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 120 python tools/dec_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/dec.log
timeout -k 10 600 python -m pytest tests/test_train_nodes_gpu.py tests/test_postproc_gpu.py tests/test_model_gpu.py -q -m gpu > $O/t.log 2>&1; tail -5 $O/t.log
timeout -k 10 300 python tools/train_step_time.py > $O/train.log 2>&1; tail -4 $O/train.log
