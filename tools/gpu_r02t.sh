cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_train_nodes_gpu.py tests/test_ddp_gpu.py tests/test_layers_gpu.py tests/test_model_gpu.py -q -m gpu > $O/t.log 2>&1; tail -30 $O/t.log
