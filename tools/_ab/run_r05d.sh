cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05d; mkdir -p $O; rm -f $O/*
timeout -k 10 300 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "winograd_f4x4" 2>&1 | tail -3
timeout -k 10 300 python tools/time_wino4_sk.py 2>&1 | grep -v amdgpu.ids | tee $O/time_sk.txt
export FD_LIB=$PWD/tools/_ab/libw4timing.so
FD_W4_TS=$O/w4_ts.txt timeout -k 10 200 python tools/time_wino4_sk.py > $O/log.txt 2>&1
awk 'NR%23==2 || NR%23==12' $O/w4_ts.txt | cut -c1-260
