set -x
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_n.log 2>&1 || { tail -20 gpurun_out/build_n.log; exit 1; }
bash tools/prof.sh r02n 2>&1 | tail -5
bash tools/prof_train.sh r02n_train 2>&1 | tail -5
timeout -k 10 200 python -m pytest tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -3
