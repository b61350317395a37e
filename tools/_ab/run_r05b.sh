cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05b; mkdir -p $O
export FD_LIB=$PWD/tools/_ab/libw4timing.so
for d in 0 8 16; do FD_W4_DBG=$d FD_W4_TS=$O/w4_ts.txt timeout -k 10 200 python tools/time_wino4_fixed.py > /dev/null 2>&1; done
grep "wgs 256\|wgs 1024" $O/w4_ts.txt
