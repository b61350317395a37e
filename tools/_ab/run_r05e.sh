cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05e; mkdir -p $O; rm -f $O/*
export FD_LIB=$PWD/tools/_ab/libw4timing.so
FD_W4_TS=$O/w4_ts.txt timeout -k 10 200 python tools/time_wino4_sk.py > $O/log.txt 2>&1
awk 'NR%23==2 || NR%23==12' $O/w4_ts.txt | cut -c1-290
