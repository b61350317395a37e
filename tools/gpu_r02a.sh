set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r02a/build.log 2>&1 || { tail -20 gpurun_out/r02a/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_effnet_gpu.py tests/test_dist_gpu.py -x -q -m gpu > gpurun_out/r02a/t_new.log 2>&1; echo "new tests rc=$?" 
tail -30 gpurun_out/r02a/t_new.log
timeout -k 10 120 python tools/grad_strides.py > gpurun_out/r02a/strides.log 2>&1; tail -15 gpurun_out/r02a/strides.log
timeout -k 10 120 python tools/grad_strides.py frozen > gpurun_out/r02a/strides_frozen.log 2>&1; tail -8 gpurun_out/r02a/strides_frozen.log
