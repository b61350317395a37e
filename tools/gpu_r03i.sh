cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
FD_AUTOTUNE=1 FD_AMP=1 timeout -k 10 900 python tools/tune_train.py 2>&1 | grep -v amdgpu | tail -3
cp gpurun_out/gfx950_tiles.json $O/gfx950_tiles_with_f16.json
cp gpurun_out/gfx950_tiles.json pytorch_object_detection_amd/tuned/gfx950_tiles.json
timeout -k 10 300 python bench.py --mode train --amp 2>/dev/null | cut -c1-200
timeout -k 10 300 python -m pytest tests/test_amp_gpu.py -m gpu -q 2>&1 | tail -2
