set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_layers_gpu.py -q -m gpu -k "patch or conv_single or conv_pyramid" > $O/t.log 2>&1; tail -15 $O/t.log
timeout -k 10 300 python tools/time_patch.py 2>&1 | grep -v amdgpu.ids | tee $O/time_patch.log
