# FD_TILE_WAVE64 against the workgroup tiles on the 1x1 layer shapes of the bench model (B = 16, 640 x 640).  -> gpurun_out/wave/
cd $GRAFT_REPO_ROOT
O=gpurun_out/wave; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_layers_gpu.py -q -x -m gpu -k "wave_tile or autotune_force" 2>&1 | tail -3
{ for a in "64 256 1 1 160 160 1" "256 64 1 1 160 160 0" "64 256 1 1 160 160 0 0" "256 128 1 1 160 160 0" "128 512 1 1 80 80 1" "512 128 1 1 80 80 0" \
           "512 256 1 1 80 80 0" "1024 256 1 1 40 40 0" "256 1024 1 1 40 40 1" "2048 512 1 1 20 20 0" "512 2048 1 1 20 20 1" "256 512 1 1 341 25 0 0" "512 256 1 1 341 25 1 0"; do
    FD_TILES=${FD_TILES:-4,8,9,7,15} timeout -k 10 100 python tools/time_conv.py $a; done; } 2>&1 | grep -v amdgpu.ids | tee $O/time_wave.txt
